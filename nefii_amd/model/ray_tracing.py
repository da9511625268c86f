"""RayTracing with the reference's constructor and forward signature (code/model/ray_tracing.py:6-101),
executed by the round-based HIP tracer (csrc/nefii_tracer.hip)."""
import os

import torch
import torch.nn as nn

from .. import ops


class RayTracing(nn.Module):
    def __init__(self, object_bounding_sphere=1.0, sdf_threshold=5.0e-5, line_search_step=0.5, line_step_iters=1,
                 sphere_tracing_iters=10, n_steps=100, n_rootfind_steps=8, trace_tier=None):
        """trace_tier (not a reference kwarg; conf key model.ray_tracer.trace_tier): the tiered sphere tracing of DESIGN 4f,
        a PER-MODEL switch - see tier_for."""
        super().__init__()
        self.object_bounding_sphere = object_bounding_sphere
        self.sdf_threshold = sdf_threshold
        self.sphere_tracing_iters = sphere_tracing_iters
        self.line_step_iters = line_step_iters
        self.line_search_step = line_search_step
        self.n_steps = n_steps
        self.n_rootfind_steps = n_rootfind_steps
        self._net = None
        # SDF arithmetic inside the tracer: 'f16x3w' (default: 3x fp16 hi/lo split MFMA, fp32-class accuracy,
        # 64-query tiles), 'f16x3' (same arithmetic, 32-query tiles) or 'f32' (f32-input MFMA, bit-exact fp32 fma)
        self.precision = os.environ.get('NEFII_TRACER_PRECISION', 'f16x3w')
        self._lin = None
        self.last_counters = None
        self.counter_sum = None       # summed over calls while collect_counters (primary + secondary traces)
        self.collect_counters = False
        self.minsdf_steps_override = None     # parity tests replay the reference's captured uniforms
        self._calls = 0
        # speculative bisection levels per round, 0 = automatic (auto_levels)
        self.bisect_levels = int(os.environ.get('NEFII_BISECT_LEVELS', '0'))
        self.concurrent = False       # this trace runs beside other work (TrainStep.prefetch_trace): throughput-bound
        self.adaptive_rounds = True     # skip the trailing empty rounds (ops.TraceRounds)
        self._rounds_state = {}
        # concurrent ray chunks on separate streams (ops.trace_rays): measured on config 2, 1/2/3/4 chunks give
        # 8.16/8.34/8.37/9.17 ms per step - the dense rounds lose what the latency rounds gain - so the default is 1
        self.stream_groups = int(os.environ.get('NEFII_TRACER_GROUPS', '0'))
        # Training mode runs minimal_sdf_points (ray_tracing.py:309-337) for the rays that miss: 67 of ~102 SDF evaluations
        # per primary ray.  With frozen geometry its results reach only `points`/`sdf_output` of miss rays and the VALUE
        # of mask_loss - no gradient (algorithmically dead).  True drops it (the tracer runs
        # its eval-mode schedule): same gradients and parameter trajectory, different mask_loss value.  Default False
        # keeps the reference's outputs.
        self.skip_min_sdf_search = False
        # with skip_min_sdf_search: still make the search's uniform draw (and drop it), so that a schedule that runs the
        # search on some iterations only (TrainStep.min_sdf_every) leaves the host RNG stream where the every-iteration
        # schedule leaves it
        self.draw_when_skipped = False
        # False while a caller traces rays whose MISS outputs nothing reads (the secondary rays of pt_render_indirect_mlp:
        # their only consumer masks them with the hit mask, idr_train.py:819): the trace then runs the reference's
        # eval-mode recurrences (ray_tracing.py:62-96 without the `if self.training` blocks: no min-SDF search for the rays
        # that leave without a hit) with nefii_tracer_params.unread_misses (ABI 14: no argmin fallback in the bracket search
        # either) - whose hits are the training path's
        self.miss_search = True
        # True while TrainStep traces several batches as ONE call (cam_loc [G,3], ray_directions [G,S,3]): every batch keeps
        # the uniform draw of the min-SDF search the reference makes per call (ray_tracing.py:316), in batch order
        self.steps_per_batch = False
        # a list: traces append their deferred round-prefix checks instead of syncing (ops.trace_rays `deferred`)
        self.deferred_checks = None
        # Coarse pass (nefii_tracer_params.coarse_tau): the 100 samples of the bracket search and of the min-SDF search are
        # evaluated in ONE fp16 pass first, and only the samples that decide a sign change / the argmin are re-evaluated
        # in split precision - same decisions and outputs as without it, ~2.5x less matrix work on ~85 % of the
        # evaluations.  The error bound is measured per network (ImplicitNetwork.coarse_tau); coarse_tau_override pins
        # it (tests), NEFII_TRACER_COARSE=0 turns the pass off.
        self.coarse = os.environ.get('NEFII_TRACER_COARSE', '1') != '0'
        self.coarse_tau_override = None
        self.coarse_cap = int(os.environ.get('NEFII_TRACER_COARSE_CAP', '0'))
        self.coarse_min_rays = int(os.environ.get('NEFII_COARSE_MIN_RAYS', '1024'))      # batches of up to this many rays: no coarse pass
        # Tiered sphere tracing (nefii_tracer_params.trace_tier; needs the coarse pass): the sphere-tracing evaluations whose
        # front is still far from the surface run on the single-pass evaluator and their value is taken as it is once it is
        # out of the band where it could decide differently.  Changes VALUES (fronts move by v16 instead of v): depths of
        # converged rays differ by up to ~sdf_threshold / cos, a handful of knife-edge rays change path (DESIGN.md 4f and the
        # error budget of section 2).  A property of the MODEL / the run, never of a call: the constructor kwarg (conf key
        # model.ray_tracer.trace_tier), overridden by NEFII_TRACE_TIER=1 / 0; default off.  Rounds 4-5 switched it by the
        # number of rays in the tracer call, which made a ray's result depend on how many other rays were traced with it
        # (world size, per-rank shard, TrainStep's trace grouping) - withdrawn.
        env = os.environ.get('NEFII_TRACE_TIER')
        self.trace_tier = (bool(trace_tier) if trace_tier is not None else self.TIER_DEFAULT) if env is None or env == '' \
            else env != '0'
        self.tier_kappa = float(os.environ.get('NEFII_TIER_KAPPA', '0'))
        self.tier_gate = float(os.environ.get('NEFII_TIER_GATE', '0'))
        # Staged searches (nefii_tracer_params.minsdf_lipschitz; need the coarse pass): a quarter of the min-SDF search's depths,
        # spread over their sorted order, first; a depth whose lower bound from its evaluated neighbours and the network's
        # measured slope bound L (ImplicitNetwork.minsdf_lipschitz) already exceeds the lowest value seen is never evaluated.
        # Likewise the bracket search of eval-mode traces (secondary rays, renders) and of rays outside the object mask: a sample
        # that the bound proves positive - and not the argmin, where the argmin matters - is never evaluated.
        # Same decisions - bit-identical outputs - provided L holds (audited by the tracer).  NEFII_MINSDF_STAGED=0 turns both off;
        # minsdf_lipschitz_override pins L (tests).
        self.minsdf_staged = os.environ.get('NEFII_MINSDF_STAGED', '1') != '0'
        # ... but NOT, by default, for the bracket search of EVAL-MODE traces (renders, the Monte-Carlo renderer's secondary rays;
        # round 6).  There a sample wrongly "proved positive" is a missed crossing - a wrong hit, not a dead loss value - and L is a
        # MEASURED bound: tests/test_gpu_kernels.py::test_tracer_staged_bracket_search_adversarial_dent builds a steep pocket of
        # radius 0.01 that the calibration's sample does not find; the online audit then fires for every trace in which >= 64 rays
        # cross the pocket (the trace is repeated without the staging) but bundles of 1-16 rays miss it silently in 3-10 % of the
        # traces.  So the eval-mode staging is an opt-in per model / run (this attribute, NEFII_BRACKET_STAGED_EVAL=1, the
        # runner's --bracket_staged_eval): bit-identical whenever L bounds the slope, config 3 -7 %, config 5 -13 %.
        self.bracket_staged_eval = os.environ.get('NEFII_BRACKET_STAGED_EVAL', '0') == '1'
        self.minsdf_lipschitz_override = None
        # The split evaluator's two correction products on block-scaled fp8 MFMAs (nefii_tracer_params.split_fp8, ABI 15):
        # a third arithmetic (|sdf error| ~1e-5 against the fp16 split's 5e-7) for every split-precision evaluation of the trace;
        # like the tier a per-MODEL switch, off by default (NEFII_SPLIT_FP8=1; 512-wide nets only, ignored elsewhere).
        self.split_fp8 = os.environ.get('NEFII_SPLIT_FP8', '0') == '1'
        self.retraced_calls = 0       # synchronous traces repeated because their online audit found a bound violated (forward)

    @staticmethod
    def auto_levels(n_rays, concurrent=False):
        """Bisection steps resolved per round (2^levels - 1 speculative evaluations per ray and round, bit-identical
        result).  A lone small batch is latency-bound: 5 levels (32 steps in 7 rounds, 31 nodes per ray and round).  A batch
        that fills the chip on its own is throughput-bound: every wasted node costs as much as a needed one, a round costs
        next to nothing - 1 level, no speculation (config 3: 308 -> 289 ms per step, config 4: 293 -> 278).  In between,
        and for traces that run beside other work, 3."""
        if n_rays >= 131072:
            return 1
        if concurrent or n_rays > 16384:
            return 3
        return 5

    @staticmethod
    def small_round_for(n_rays, concurrent):
        """Largest round (split-precision queries) that runs on 32-query tiles.  A lone trace wants short rounds: 8192
        (one 32-query tile per CU).  A trace that runs beside others (TrainStep's lookahead) is bound by chip time, and a
        32-query tile costs 1.5x the chip time per query of a 64-query one: only rounds that would not even give a
        quarter of the CUs a 64-query tile stay on the small tiles.  NEFII_SMALL_ROUND overrides."""
        env = os.environ.get('NEFII_SMALL_ROUND')
        if env:
            return int(env)
        return 4096 if concurrent and n_rays > 1024 else 0

    # Off unless the model / the run asks for it (the runner's --trace_tier, conf model.ray_tracer.trace_tier, bench.py,
    # NEFII_TRACE_TIER=1): the untiered trace is the one whose every decision and value is the split evaluator's.  What the
    # tier buys where evaluations, not round latency, make the trace: config 3 169 -> 143 ms per step, config 4 126 -> 115,
    # config 2 2.05 -> 1.92 (round 5) - after the parity protocol of DESIGN.md section 4f.
    TIER_DEFAULT = False

    def tier_for(self, n_rays=None):
        """Whether traces of this model take the tier.  n_rays is ignored (kept for callers of the rounds-4/5 rule, which
        switched at 32768 rays per call): a ray's result does not depend on the size of the call it is traced in."""
        return bool(self.trace_tier)

    def bind(self, implicit_network):
        """The kernels evaluate the SDF MLP themselves, so the tracer needs the network, not a closure."""
        object.__setattr__(self, '_net', implicit_network)

    def _cfg(self):
        return dict(object_bounding_sphere=self.object_bounding_sphere, sdf_threshold=self.sdf_threshold,
                    line_search_step=self.line_search_step, line_step_iters=self.line_step_iters,
                    sphere_tracing_iters=self.sphere_tracing_iters, n_steps=self.n_steps,
                    n_rootfind_steps=self.n_rootfind_steps)

    def _bounds(self, net, n_rays, frozen, training=True):
        """(coarse_tau, minsdf_lipschitz, audit callable or None) for a call of n_rays rays with the network's CURRENT bounds."""
        tau, lip, audit = 0.0, 0.0, None
        # (batches of up to 1024 rays are latency-bound - a handful of tiles per round: the coarse pass's extra round per
        # dense search costs them more than its cheaper samples save; config 1: 2.39 vs 2.2 ms per step)
        # ... and geometry that still trains (model/trainable_geometry.py) changes its weights every step: the bound would
        # have to be re-measured per forward (two 65 k-point evaluations and a host sync) - no coarse pass there
        # (the tier needs the coarse pass's bound and evaluator: a model that takes the tier runs the pass whatever the size of
        # the call, so that its arithmetic does not change with the batch)
        if self.coarse and self.precision == 'f16x3w' and (n_rays > self.coarse_min_rays or self.tier_for()) and \
                (frozen or self.coarse_tau_override is not None):
            tau = self.coarse_tau_override if self.coarse_tau_override is not None else \
                net.coarse_tau(self.object_bounding_sphere)
            if tau > 0 and self.minsdf_staged and (training or self.bracket_staged_eval):
                lip = self.minsdf_lipschitz_override if self.minsdf_lipschitz_override is not None else \
                    net.minsdf_lipschitz(self.object_bounding_sphere)
            if tau > 0 and (self.coarse_tau_override is None or (lip > 0 and self.minsdf_lipschitz_override is None)):
                # every refined sample is evaluated both ways: the tracer reports the largest difference it saw and the
                # network compares it with the bound it claimed (ImplicitNetwork.note_coarse_audit); likewise for the
                # slope bound of the staged min-SDF search (note_lipschitz_audit)
                radius, used, lip_used = self.object_bounding_sphere, tau, lip
                check_tau, check_lip = self.coarse_tau_override is None, self.minsdf_lipschitz_override is None

                def audit(v, lip_violation=0.0):
                    if check_tau:
                        net.note_coarse_audit(v, used, radius)
                    # (the slope check presumes |coarse - exact| < tau: a trace whose tau audit failed says nothing about L -
                    # and without the coarse pass there is no staged search to switch off)
                    if check_lip and not v > used:
                        net.note_lipschitz_audit(lip_violation, lip_used)
        return tau, lip, audit

    def forward(self, sdf, cam_loc, object_mask, ray_directions):
        """cam_loc [B,3], ray_directions [B,S,3], object_mask [B*S] -> (points [B*S,3], hit [B*S], dists [B*S]).

        ``sdf``: the reference passes ``lambda x: implicit_network(x)[:, 0]``.  A Python closure cannot run
        inside a kernel, so the tracer uses the ImplicitNetwork it was bound to (IDRNetwork binds it); passing
        the ImplicitNetwork itself as ``sdf`` also works.  Any other callable without a bound network raises."""
        net = sdf if hasattr(sdf, 'packed') else self._net
        if net is None:
            raise RuntimeError('RayTracing: no ImplicitNetwork bound; call ray_tracer.bind(implicit_network) or '
                               'pass the network as `sdf`')
        B, S, _ = ray_directions.shape
        dirs = ops._f32(ray_directions).reshape(-1, 3)
        origins = ops._f32(cam_loc).unsqueeze(1).expand(B, S, 3).reshape(-1, 3).contiguous()
        dev = dirs.device
        if self._lin is None or self._lin.device != dev or self._lin.numel() != self.n_steps:
            self._lin = torch.linspace(0, 1, steps=self.n_steps).to(dev)     # ray_tracing.py:203
        steps = None
        training = self.training and not self.skip_min_sdf_search and self.miss_search
        if self.training and not self.skip_min_sdf_search and not self.miss_search and \
                isinstance(self.minsdf_steps_override, (list, tuple)):
            self._calls += 1        # this call's entry of the per-call override list stays unused
        if self.training and self.skip_min_sdf_search and self.draw_when_skipped and self.miss_search and \
                self.minsdf_steps_override is None:
            for _ in range(B if (self.steps_per_batch and B > 1) else 1):
                torch.empty(self.n_steps).uniform_(0.0, 1.0)
        group = 0
        if training:
            rows = B if (self.steps_per_batch and B > 1) else 1
            drawn = []
            for _ in range(rows):
                if self.minsdf_steps_override is not None:
                    ov = self.minsdf_steps_override
                    if isinstance(ov, (list, tuple)):      # one entry per trace call (primary, secondary, ...)
                        ov = ov[self._calls % len(ov)]
                        self._calls += 1
                    drawn.append(ov.reshape(-1))
                else:
                    # drawn on the host exactly like minimal_sdf_points (:316); always drawn (the reference draws only
                    # when some ray needs the search, a data-dependent host sync this build avoids)
                    drawn.append(torch.empty(self.n_steps).uniform_(0.0, 1.0))
            steps = drawn[0] if rows == 1 else torch.stack([d.cpu() for d in drawn])
            if steps.device.type == 'cpu' and dev.type == 'cuda':
                # pinned + non_blocking: a pageable host-to-device copy blocks the HOST until it has executed, and on a trace
                # stream it executes behind the trace(s) enqueued before it
                steps = steps.contiguous().pin_memory().to(dev, non_blocking=True)
            else:
                steps = steps.to(dev).contiguous()
            group = S if rows > 1 else 0
        n_rays = dirs.shape[0]
        levels = self.bisect_levels or self.auto_levels(n_rays, self.concurrent)
        frozen = not any(p.requires_grad for p in ops.param_list(net))
        state = None
        if self.adaptive_rounds:      # one guess per (mode, batch size): primary and secondary traces differ
            import math
            state = self._rounds_state.setdefault((training, int(math.log2(n_rays + 1))), ops.TraceRounds())
        deferred = self.deferred_checks
        # A trace whose online audit finds its coarse bound tau or its slope bound L violated has made decisions on the strength of
        # a bound that does not hold (a skipped sample may have been the crossing / the argmin): note_*_audit switches the
        # pass off for these weights, and THIS trace is repeated without it - same rays, same min-SDF draws - before anything
        # reads its result.  Synchronous callers (a step's own trace, model(input), eval renders, the secondary trace) get that
        # here; traces enqueued ahead report through their deferred checks and TrainStep._take_prefetched re-traces them.
        for attempt in range(3):
            seen = len(net.coarse_audit_events)
            tau, lip, audit = self._bounds(net, n_rays, frozen, training)
            params = ops.make_tracer_params(self._cfg(), training, self.precision, levels, coarse_tau=tau,
                                            coarse_cap=self.coarse_cap, minsdf_group=group,
                                            small_round=self.small_round_for(n_rays, self.concurrent),
                                            trace_tier=self.tier_for(), tier_kappa=self.tier_kappa,
                                            tier_gate=self.tier_gate, minsdf_lipschitz=lip,
                                            unread_misses=0 if self.miss_search else 1, split_fp8=self.split_fp8)
            res = ops.trace_rays(net.packed(f16x3=self.precision.startswith('f16x3')), params, origins, dirs,
                                 object_mask.reshape(-1), self._lin, steps, want_counters=self.collect_counters,
                                 rounds_state=state,
                                 groups=1 if group else (self.stream_groups or 1),
                                 deferred=deferred,
                                 audit=audit)
            if deferred is not None or audit is None or not any(
                    kind in ('disabled', 'lipschitz_disabled') for kind, _, _ in net.coarse_audit_events[seen:]):
                break
            self.retraced_calls += 1
        if self.collect_counters:
            self.last_counters = res[3]
            cur = res[3]
            if self.counter_sum is not None and self.counter_sum.shape[0] != cur.shape[0]:
                # traces of one step differ in their round budget (bisection levels by batch size, coarse pass or not)
                n = max(self.counter_sum.shape[0], cur.shape[0])
                pad = lambda t: torch.cat([t, t.new_zeros(n - t.shape[0], t.shape[1])]) if t.shape[0] < n else t
                self.counter_sum, cur = pad(self.counter_sum), pad(cur)
            self.counter_sum = cur.clone() if self.counter_sum is None else ops.sum_counters(self.counter_sum, cur)
        return res[0], res[1], res[2]
