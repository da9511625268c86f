"""IDRNetwork / ImplicitNetwork / RenderingNetwork with the reference's call surface, on HIP kernels.

Mirrors code/model/implicit_differentiable_renderer.py of FuxiComputerVision/Nefii:
  ImplicitNetwork  :18-123    RenderingNetwork :126-241    IDRNetwork :244-759
Same constructor kwargs (conf blocks), same state-dict keys (weight-norm ``lin{l}.weight_g/.weight_v/.bias``),
same ``forward(input, with_point=False)`` contract and output dict (:460-477), so a conf with
``train.model_class = nefii_amd.model.implicit_differentiable_renderer.IDRNetwork`` swaps the implementation
(utils.get_class hook, general.py:10-16).

What runs where: every numeric stage of the path - camera rays, sphere tracing, SDF value/normal, the
radiance and material MLPs, SG shading - is a hand-written gfx950 kernel in libnefii_hip.so (no eager
fallback; ops raise without the library).  torch is used for parameter storage, the weight-norm
reparameterisation (tiny [out,in] tensors), masking/scatter glue and autograd bookkeeping.

Scope: Step-2 with frozen geometry (``freeze_geometry()``; every shipped Step-2 script passes
--freeze_geometry) on the kernels.  The trainable-geometry branch (:357-393, SampleNetwork) exists as a torch slow path
for the closed-form ``sg`` render type (model/trainable_geometry.py).
"""
import math
import os

import numpy as np
import torch
import torch.nn as nn

from .. import ops
from ..utils import rend_util
from .ray_tracing import RayTracing
from .sample_network import SampleNetwork
from .sg_envmap_material import EnvmapMaterialNetwork
from .sg_render import render_with_sg


def _params_version(module):
    """Changes whenever a parameter's storage or content does (ops.param_list: the parameter list is gathered once)."""
    return tuple((p.data_ptr(), p._version) for p in _plist(module))


_plist = ops.param_list


class ImplicitNetwork(nn.Module):
    """SDF MLP (+ feature vector).  forward() -> [N, 1+F] and gradient() -> [N,1,3] run as fused kernels."""

    def __init__(self, feature_vector_size, d_in, d_out, dims, geometric_init=True, bias=1.0, skip_in=(),
                 weight_norm=True, multires=0, use_last_as_f=False):
        super().__init__()
        if use_last_as_f:
            assert feature_vector_size == dims[-1]
        assert d_in == 3
        self.feature_vector_size = feature_vector_size
        self.cfg = dict(d_in=d_in, d_out=d_out, dims=list(dims), skip_in=list(skip_in), multires=multires,
                        use_last_as_f=use_last_as_f)
        self.specs, self.enc = ops.sdf_specs(self.cfg, feature_vector_size)
        self.num_layers = len(self.specs) + 1
        self.skip_in = skip_in
        self.use_last_as_f = use_last_as_f
        self.weight_norm = weight_norm
        d0 = self.specs[0].k_in
        for l, s in enumerate(self.specs):
            lin = nn.Linear(s.k_in, s.n_out)
            if geometric_init:      # same statistics as the reference's geometric initialisation (:62-76)
                if l == len(self.specs) - 1:
                    nn.init.normal_(lin.weight, mean=np.sqrt(np.pi) / np.sqrt(s.k_in), std=0.0001)
                    nn.init.constant_(lin.bias, -bias)
                else:
                    nn.init.constant_(lin.bias, 0.0)
                    nn.init.normal_(lin.weight, 0.0, np.sqrt(2) / np.sqrt(s.n_out))
                    if multires > 0 and l == 0:
                        nn.init.constant_(lin.weight[:, 3:], 0.0)
                    elif multires > 0 and l in skip_in:
                        nn.init.constant_(lin.weight[:, -(d0 - 3):], 0.0)
            if weight_norm:
                lin = nn.utils.weight_norm(lin)
            setattr(self, 'lin' + str(l), lin)
        self._pm = None
        self._pm_version = None
        self._pm_fit = None
        self._tau = None          # (packed version, radius, error bound of the tracer's coarse pass for these weights)
        self.coarse_audit_max = 0.0       # largest |coarse - split| the tracer has reported for a refined sample
        self.coarse_audit_events = []     # ('recalibrated' | 'disabled' | 'lipschitz_disabled', observed, bound in use): what note_coarse_audit did
        self._lip = None                  # (packed version, radius, L): Lipschitz bound for the tracer's staged min-SDF search

    def coarse_tau(self, radius=1.0):
        """Error bound of the tracer's single-pass evaluator for the current weights (ops.calibrate_coarse_tau), measured
        once per packed version - i.e. once, with frozen geometry.  0.0 if the net has no single-pass stream."""
        pm = self.packed(f16x3=True)
        if self._tau is None or self._tau[0] != self._pm_version or self._tau[1] != radius:
            self._tau = (self._pm_version, radius, ops.calibrate_coarse_tau(pm, radius))
        return self._tau[2]

    def minsdf_lipschitz(self, radius=1.0):
        """Bound on the SDF's slope along a ray for the current weights (ops.calibrate_lipschitz: 1.5 x the largest |grad sdf|
        found in the bounding sphere by a random sample and a local search around its steepest points), measured once per
        packed version; 0.0 after the tracer's audit has seen it fail (note_lipschitz_audit)."""
        self.packed(f16x3=True)
        if self._lip is None or self._lip[0] != self._pm_version or self._lip[1] != radius:
            with torch.no_grad():
                self._lip = (self._pm_version, radius, ops.calibrate_lipschitz(self.gradient, _plist(self)[0].device, radius))
        return self._lip[2]

    def note_lipschitz_audit(self, violation, lip_used):
        """The tracer's report on nefii_tracer_params.minsdf_lipschitz for one trace (counter 12): the largest amount by which
        a depth of a staged search's second stage fell below the lower bound that kept it.  Above 0 the claimed constant
        does not hold for these weights - a depth that was skipped might have been the argmin - so the staged search is
        switched off for them (every depth is evaluated again), with a warning."""
        if violation <= 0.0 or lip_used <= 0.0 or self._lip is None or self._lip[0] != self._pm_version or self._lip[2] <= 0.0:
            return
        import warnings
        warnings.warn('staged min-SDF search of the tracer disabled for this network: a depth lay %.3e below the lower bound '
                      'its claimed Lipschitz constant %.3f gives' % (violation, lip_used))
        self._lip = (self._pm_version, self._lip[1], 0.0)
        self.coarse_audit_events.append(('lipschitz_disabled', float(violation), float(lip_used)))

    def note_coarse_audit(self, observed, tau_used, radius=1.0):
        """The tracer's report for one trace: the largest |single pass - split| among the coarse samples it re-evaluated
        (every refined sample is evaluated both ways - nefii_trace_rays, counter 8).  coarse_tau is a MEASURED bound (3 x the
        largest difference over 65 536 random points: ops.calibrate_coarse_tau), not a proven one: a trained network with
        sharper features than the calibration sample saw could exceed it.  observed > bound / 2 (the margin has shrunk below 2;
        tools/tau_probe.py saw 2.2-2.5 over 16.8 M points): the bound is raised to 3 x observed.  observed > bound: a refined
        sample was further from its exact
        value than the decisions of that trace assumed - a sample that was NOT refined may have decided differently - so the
        coarse pass is switched off for these weights (bound 0: every sample in split precision) with a warning."""
        self.coarse_audit_max = max(self.coarse_audit_max, float(observed))
        if self._tau is None or self._tau[0] != self._pm_version or self._tau[2] <= 0.0 or tau_used <= 0.0:
            return
        tau = self._tau[2]
        if observed > tau_used:
            import warnings
            warnings.warn('coarse pass of the tracer disabled for this network: a refined sample differed from its '
                          'single-pass value by %.3e, above the claimed bound %.3e' % (observed, tau_used))
            self._tau = (self._pm_version, self._tau[1], 0.0)
            self.coarse_audit_events.append(('disabled', float(observed), float(tau_used)))
        elif observed * 2.0 > tau:
            self._tau = (self._pm_version, self._tau[1], ops.COARSE_TAU_SAFETY * float(observed))
            self.coarse_audit_events.append(('recalibrated', float(observed), float(tau_used)))

    def effective_weights(self):
        ws, bs = [], []
        for l in range(len(self.specs)):
            lin = getattr(self, 'lin' + str(l))
            if self.weight_norm:
                ws.append(torch._weight_norm(lin.weight_v, lin.weight_g, 0))
            else:
                ws.append(lin.weight)
            bs.append(lin.bias)
        return ws, bs

    def packed(self, f16x3=False):
        """Packed MFMA-order weights; repacked only when a parameter changed (never, once geometry is frozen)."""
        ver = _params_version(self)
        dev = _plist(self)[0].device
        if self._pm is None or self._pm.device != dev or (f16x3 and not self._pm.f16x3):
            self._pm = ops.PackedMLP(self.specs, ops.ACT_SOFTPLUS100, ops.HEAD_NONE, self.enc, 0, dev, f16x3=f16x3)
            self._pm_version = None
        if self._pm_version != ver:
            with torch.no_grad():
                ws, bs = self.effective_weights()
                self._pm.pack(ws, bs)
            self._pm_version = ver
        return self._pm

    def _check_frozen(self):
        if torch.is_grad_enabled() and any(p.requires_grad for p in _plist(self)):
            raise NotImplementedError(
                'nefii_amd: differentiating through the SDF network (trainable geometry) is outside the Step-2 '
                'hot path; call IDRNetwork.freeze_geometry() as every shipped Step-2 script does')

    def _trainable(self):
        return torch.is_grad_enabled() and any(p.requires_grad for p in _plist(self))

    def forward(self, input, compute_grad=False):
        x = ops._f32(input)
        if self._trainable():
            # Step-1 geometry fit (geometry_train.py:361): values at given positions, differentiable wrt the weights
            # (fused forward / backward / weight-gradient kernels; weight_norm stays in autograd).  Gradients wrt the
            # positions - what trainable geometry inside the renderer would need - are not provided: gradient() and the
            # tracer keep refusing unfrozen parameters.
            ws, bs = self.effective_weights()
            pm = ops.PackedMLP(self.specs, ops.ACT_SOFTPLUS100, ops.HEAD_NONE, self.enc, 0, x.device) \
                if self._pm_fit is None or self._pm_fit.device != x.device else self._pm_fit
            self._pm_fit = pm
            if self.use_last_as_f:      # the feature columns (last hidden activation) come back as constants
                out, hidden = ops.FusedMLPHiddenFn.apply(pm, x.detach(), *ws, *bs)
                return torch.cat([out, hidden], dim=-1)
            return ops.FusedMLPFn.apply(pm, x.detach(), None, None, None, *ws, *bs)
        out, hidden, _ = ops.mlp_forward(self.packed(), x, None, None, None, want_hidden=self.use_last_as_f)
        if self.use_last_as_f:
            out = torch.cat([out, hidden], dim=-1)
        return out

    def gradient(self, x, no_grad=False):
        self._check_frozen()
        _, _, g = ops.sdf_value_grad(self.packed(), ops._f32(x))
        return g.unsqueeze(1)

    def value_feature_gradient(self, x):
        """One pass for what get_rbg_value needs: (sdf [N,1], feature [N,F] or None, d sdf/dx [N,3])."""
        self._check_frozen()
        out, hidden, g = ops.sdf_value_grad(self.packed(), ops._f32(x), want_feat=self.use_last_as_f)
        if self.use_last_as_f:
            feat = hidden
        elif self.feature_vector_size > 0:
            feat = out[:, 1:].contiguous()
        else:
            feat = None
        return out[:, :1], feat, g


class RenderingNetwork(nn.Module):
    """View-dependent radiance MLP: cat[PE(x), PE(v), n, feat] -> ReLU MLP -> head, one fused kernel."""

    def __init__(self, feature_vector_size, mode, d_in, d_out, dims, weight_norm=True, weight_init=False,
                 multires_view=0, multires_xyz=0, normalize_output=True, clip_output=False, clip_method='relu'):
        super().__init__()
        self.feature_vector_size = feature_vector_size
        self.mode = mode
        self.normalize_output = normalize_output
        self.clip_output = clip_output
        self.clip_method = clip_method
        self.weight_norm = weight_norm
        self.cfg = dict(mode=mode, d_in=d_in, d_out=d_out, dims=list(dims), multires_view=multires_view,
                        multires_xyz=multires_xyz, normalize_output=normalize_output, clip_output=clip_output,
                        clip_method=clip_method)
        self.specs, self.enc, self.head = ops.radiance_specs(self.cfg, feature_vector_size)
        self.num_layers = len(self.specs) + 1
        lins = []
        for l, s in enumerate(self.specs):
            lins.append(nn.Linear(s.k_in, s.n_out))
        if weight_init:             # :179-191
            for lin in lins[:-1]:
                nn.init.kaiming_uniform_(lin.weight, mode='fan_in', nonlinearity='relu')
                nn.init.constant_(lin.bias, 0.0)
            nn.init.constant_(lins[-1].bias, 0.0)
            if normalize_output:
                nn.init.xavier_uniform_(lins[-1].weight, gain=nn.init.calculate_gain('tanh'))
            elif clip_method == 'relu':
                nn.init.kaiming_uniform_(lins[-1].weight, mode='fan_in', nonlinearity='relu')
        for l, lin in enumerate(lins):
            if weight_norm:
                lin = nn.utils.weight_norm(lin)
            setattr(self, 'lin' + str(l), lin)
        self._pm = None
        # True: nothing differentiates this network's output in the current run (a training run whose loss weights the
        # radiance colour with 0 - physg.conf - detaches it: training/step.py sets this) - forward then takes the no-grad path
        self.outputs_detached = False
        self._packed_for = None

    def packed(self, device):
        half = ops.mlp_precision() if ops.mlp_precision() in ('f16', 'f16x3') else False
        if self._pm is None or self._pm.device != device or self._pm.half != half:
            self._pm = ops.PackedMLP(self.specs, ops.ACT_RELU, self.head, self.enc, self.feature_vector_size, device,
                                     half=half)
        return self._pm

    def _effective_weights(self):
        ws, bs = [], []
        for l in range(len(self.specs)):
            lin = getattr(self, 'lin' + str(l))
            ws.append(torch._weight_norm(lin.weight_v, lin.weight_g, 0) if self.weight_norm else lin.weight)
            bs.append(lin.bias)
        return ws, bs

    def forward(self, points, normals, view_dirs, feature_vectors=None):
        p, n, v = ops._f32(points), ops._f32(normals), ops._f32(view_dirs)
        if self.mode == 'idr':
            a, b, c = p, v, n
        elif self.mode == 'no_view_dir':
            a, b, c = p, n, None
        else:
            a, b, c = p, v, None
        feat = ops._f32(feature_vectors) if self.feature_vector_size > 0 else None
        pm = self.packed(p.device)
        if self.outputs_detached or not torch.is_grad_enabled() or not any(q.requires_grad for q in _plist(self)):
            # no gradient can reach the weights: weight norm and packing only when a parameter changed (never while the loss
            # does not train this network; once per frame chunk in a render before) - 8 launches less per call
            key = (_params_version(self), id(pm))
            with torch.no_grad():
                if self._packed_for != key:
                    pm.pack(*self._effective_weights())
                    self._packed_for = key
                return ops.mlp_forward(pm, a, b, c, feat, want_stash=False)[0]
        self._packed_for = None
        ws, bs = self._effective_weights()
        return ops.FusedMLPFn.apply(pm, a, b, c, feat, *ws, *bs)


class IDRNetwork(nn.Module):
    def __init__(self, conf):
        super().__init__()
        self.feature_vector_size = conf.get_int('feature_vector_size')
        self.correct_normal = conf.get_bool('correct_normal', default=False)
        if self.correct_normal:
            raise NotImplementedError('correct_normal is broken in the reference (attribute shadows the method)')
        self.implicit_network = ImplicitNetwork(self.feature_vector_size, **conf.get_config('implicit_network'))
        self.rendering_network = RenderingNetwork(self.feature_vector_size, **conf.get_config('rendering_network'))
        self.envmap_material_network = EnvmapMaterialNetwork(correct_normal=False,
                                                             feature_vector_size=self.feature_vector_size,
                                                             **conf.get_config('envmap_material_network'))
        self.ray_tracer = RayTracing(**conf.get_config('ray_tracer'))
        self.ray_tracer.bind(self.implicit_network)
        self.sample_network = SampleNetwork()
        self.object_bounding_sphere = conf.get_float('ray_tracer.object_bounding_sphere')
        self.render_type = conf.get_string('render_type', default='sg')
        self.rgb_render = self.get_rgb_render(self.render_type)
        self.fast_multi_ray = conf.get_bool('fast_multi_ray', default=False)
        if self.fast_multi_ray:
            raise NotImplementedError('fast_multi_ray (off in every shipped conf)')
        self.render_background = conf.get_bool('render_background', default=False)
        self.state_freeze_geo = False
        self.state_freeze_idr = False
        self.state_freeze_env_mat = False
        # True: the secondary trace also fills the outputs of rays that miss (min-SDF search, as the reference executes
        # it; nothing reads them - model/path_tracing_render.py)
        self.secondary_miss_search = os.environ.get('NEFII_SECONDARY_MISS_SEARCH', '0') == '1'

    # ---- freeze surface used by the runners (idr_train.py:621-630) ------------------------------
    def freeze_geometry(self):
        for p in self.implicit_network.parameters():
            p.requires_grad = False
        self.state_freeze_geo = True

    def unfreeze_geometry(self):
        for p in self.implicit_network.parameters():
            p.requires_grad = True
        self.state_freeze_geo = False

    def freeze_idr(self):
        self.freeze_geometry()
        for p in self.rendering_network.parameters():
            p.requires_grad = False
        self.state_freeze_idr = True

    def unfreeze_idr(self):
        self.unfreeze_geometry()
        for p in self.rendering_network.parameters():
            p.requires_grad = True
        self.state_freeze_idr = False

    def freeze_decompose_render(self):
        for p in self.envmap_material_network.parameters():
            p.requires_grad = False
        self.state_freeze_env_mat = True

    def unfreeze_decompose_render(self):
        for p in self.envmap_material_network.parameters():
            p.requires_grad = True
        self.state_freeze_env_mat = False

    def train(self, mode: bool = True):
        nn.Module.train(self, mode)
        if self.state_freeze_idr:
            self.rendering_network.eval()
        if self.state_freeze_geo:
            self.implicit_network.eval()
        if self.state_freeze_env_mat:
            self.envmap_material_network.eval()
        return self

    def forward(self, input, with_point=False):
        if not with_point:
            return self.forward_with_uv(input)
        return self.forward_with_point(input)

    # ---- forward_with_uv (:312-501) ------------------------------------------------------------------
    def forward_with_uv(self, input):
        if self.training and not self.state_freeze_geo:
            # trainable geometry (:357-393, SampleNetwork): torch slow path, closed-form shading only
            from . import trainable_geometry
            return trainable_geometry.forward_with_uv(self, input)
        ctx = self.trace_head(input)
        idx = torch.nonzero(ctx['network_object_mask']).flatten()          # one host sync per call (compaction size)
        return self.shade_tail(ctx, idx)

    def trace_head(self, input):
        """Camera rays -> surface points (no autograd): everything of forward_with_uv (:312-356) ahead of the
        compaction of the hit rays.  Returns the tensors shade_tail needs, all of the static shape [B*S*R, .]."""
        return self.attach_surface(self.trace_points(input))

    def trace_points(self, input):
        """trace_head without the SDF value / feature / gradient pass at the traced points (attach_surface)."""
        if self.training and not self.state_freeze_geo:
            raise NotImplementedError('the kernel path needs frozen geometry (freeze_geometry()); forward() routes '
                                      'trainable geometry to model/trainable_geometry.py')
        intrinsics = input['intrinsics']
        uv = input['uv']
        pose = input['pose']
        object_mask = input['object_mask'].reshape(-1)
        multi = uv.dim() == 4
        shape = None
        if multi:
            B, S, R, _ = uv.shape
            shape = (B, S, R)
            uv = uv.reshape(B, S * R, 2)
            object_mask = object_mask.reshape(B, S, 1).expand(B, S, R).reshape(-1)
        ray_dirs, cam_loc = rend_util.get_camera_params(uv, pose, intrinsics)
        with torch.no_grad():
            points, network_object_mask, dists = self.ray_tracer(sdf=self.implicit_network, cam_loc=cam_loc,
                                                                 object_mask=object_mask, ray_directions=ray_dirs)
        return {'points': points, 'network_object_mask': network_object_mask, 'object_mask': object_mask,
                'ray_dirs': ray_dirs.reshape(-1, 3), 'multi': shape}

    def trace_points_group(self, inputs):
        """trace_points of several batches (one image each, equal shapes) as ONE tracer call: G x S rays per round instead
        of G traces whose late rounds each keep a handful of tiles busy.  Every batch keeps its own camera and its own
        draw of the min-SDF search (RayTracing.steps_per_batch), so each slice equals that batch's own trace.  Returns the
        per-batch ctx dicts (views into the call's outputs)."""
        if self.training and not self.state_freeze_geo:
            raise NotImplementedError('the kernel path needs frozen geometry (freeze_geometry())')
        dirs, cams, masks, shapes = [], [], [], []
        for input in inputs:
            uv, object_mask = input['uv'], input['object_mask'].reshape(-1)
            shape = None
            if uv.dim() == 4:
                B, S, R, _ = uv.shape
                shape = (B, S, R)
                uv = uv.reshape(B, S * R, 2)
                object_mask = object_mask.reshape(B, S, 1).expand(B, S, R).reshape(-1)
            ray_dirs, cam_loc = rend_util.get_camera_params(uv, input['pose'], input['intrinsics'])
            if ray_dirs.shape[0] != 1 or (dirs and ray_dirs.shape != dirs[0].shape):
                raise ValueError('trace_points_group: one image per batch, equal ray counts')
            dirs.append(ray_dirs), cams.append(cam_loc), masks.append(object_mask), shapes.append(shape)
        rt = self.ray_tracer
        rt.steps_per_batch = True
        try:
            with torch.no_grad():
                points, hit, dists = rt(sdf=self.implicit_network, cam_loc=torch.cat(cams), object_mask=torch.cat(masks),
                                        ray_directions=torch.cat(dirs))
        finally:
            rt.steps_per_batch = False
        S = dirs[0].shape[1]
        return [{'points': points[g * S:(g + 1) * S], 'network_object_mask': hit[g * S:(g + 1) * S],
                 'object_mask': masks[g], 'ray_dirs': dirs[g].reshape(-1, 3), 'multi': shapes[g]}
                for g in range(len(inputs))]

    def attach_surface(self, ctx):
        """SDF value (and, for small batches, features and gradient) at the traced points - the part of trace_head that
        has to be redone when a trace enqueued ahead of time turns out to need more rounds (training/step.py)."""
        points = ctx['points']
        with torch.no_grad():
            # the tracer already returns cam + dist * dir (reference recomputes it, :352).
            # Small batches (<= one 32-point tile per CU) are latency-bound: one value+feature+gradient pass over ALL
            # rays costs the same as the pass over the hits alone and replaces the separate SDF forward.
            pre = None
            if points.shape[0] <= int(os.environ.get('NEFII_SURFACE_ALL_MAX', str(32 * 256))):
                pre = self.implicit_network.value_feature_gradient(points)
                sdf_output = pre[0]
            else:
                # only the SDF column is needed here (the mask term of the loss): the tracer's split-precision tile
                # evaluator computes it ~4x faster than the fp32 forward that also produces the feature columns
                pm = self.implicit_network.packed(f16x3=True) if self.ray_tracer.precision.startswith('f16x3') else None
                if pm is not None and pm.f16x3:
                    sdf_output = ops.sdf_eval(pm, ops._f32(points)).unsqueeze(1)
                else:
                    sdf_output = self.implicit_network(points)[:, 0:1]
        ctx['sdf_output'], ctx['pre'] = sdf_output, pre
        return ctx

    def shade_tail(self, ctx, idx, dst=None):
        """Shading of the compacted hit rays `idx` and assembly of the output dict (:358-501).
        dst (optional, same length as idx): row of the output each entry is scattered to.  A caller that pads idx to
        a fixed length (graph capture, training/step.py) points the padding entries at row n_all, a scratch row that
        is sliced off again, and gets shapes that do not depend on the hit count; background colours are then
        evaluated for all rays and blended by mask instead of gathered."""
        points, network_object_mask, object_mask = ctx['points'], ctx['network_object_mask'], ctx['object_mask']
        sdf_output, pre, ray_dirs = ctx['sdf_output'], ctx['pre'], ctx['ray_dirs']
        surface_mask = network_object_mask
        self.last_ray_hit = network_object_mask     # per-RAY mask of this forward (the output dict's is per pixel); diagnostics
        n_all = points.shape[0]
        rows = n_all + (1 if dst is not None else 0)
        dev = points.device
        # (output buffer, get_rbg_value's key, columns, value of the rays without a hit: :441-448)
        layout = (('idr_rgb_values', 'idr_rgb', 3, 1.0), ('sg_rgb_values', 'sg_rgb', 3, 1.0), ('normal_values', 'normals', 3, 1.0),
                  ('sg_diffuse_rgb_values', 'sg_diffuse_rgb', 3, 1.0), ('sg_diffuse_albedo_values', 'sg_diffuse_albedo', 3, 1.0),
                  ('sg_specular_rgb_values', 'sg_specular_rgb', 3, 0.0), ('sg_roughness_values', 'sg_roughness', 1, 0.0),
                  ('sg_specular_reflection_values', 'sg_specular_reflectance', 3, 0.0))
        ret = {}
        if idx.numel() > 0:
            if pre is not None and os.environ.get('NEFII_PREPARE_HITS', '1') != '0':
                # the gathers of the hit rays and both normalisations in one launch (ops.prepare_hits) instead of ten eager ops
                with torch.no_grad():
                    p_h, v_h, n_h, f_h = ops.prepare_hits(points, ray_dirs, pre[2], pre[1], idx)
                ret = self.get_rbg_value(p_h, v_h, surface=(None, f_h, None), unit=(n_h, v_h))
            else:
                if pre is not None:
                    pre = (None, pre[1].index_select(0, idx) if pre[1] is not None else None, pre[2].index_select(0, idx))
                ret = self.get_rbg_value(points.index_select(0, idx), -ray_dirs.index_select(0, idx), surface=pre)
            where = idx if dst is None else dst
            # all eight buffers in two launches (+ one for all their gradients) instead of fill / expand / index_put each
            bufs = ops.assemble_rows(where, rows, [f for _, _, _, f in layout], [c for _, _, c, _ in layout],
                                     [ret[k] for _, k, _, _ in layout])
            out = {name: b for (name, _, _, _), b in zip(layout, bufs)}
        else:
            out = {name: torch.full((rows, c), f, device=dev) for name, _, c, f in layout}
        if dst is not None:
            out = {k: v[:n_all] for k, v in out.items()}
        if self.render_background:
            if dst is None:
                bidx = torch.nonzero(~surface_mask).flatten()
                if bidx.numel() > 0:
                    bg = self.get_background_rgb(ray_dirs.index_select(0, bidx))
                    out['sg_rgb_values'] = out['sg_rgb_values'].index_put((bidx,), bg)
            else:
                out['sg_rgb_values'] = torch.where(surface_mask.unsqueeze(-1), out['sg_rgb_values'],
                                                   self.get_background_rgb(ray_dirs))
        output = {
            'points': points,
            'idr_rgb_values': out['idr_rgb_values'],
            'sg_rgb_values': out['sg_rgb_values'],
            'normal_values': out['normal_values'],
            'sdf_output': sdf_output,
            'network_object_mask': network_object_mask,
            'object_mask': object_mask,
            'grad_theta': None,
            'sg_diffuse_rgb_values': out['sg_diffuse_rgb_values'],
            'sg_diffuse_albedo_values': out['sg_diffuse_albedo_values'],
            'sg_specular_rgb_values': out['sg_specular_rgb_values'],
            'sg_roughness_values': out['sg_roughness_values'],
            'sg_specular_reflection_values': out['sg_specular_reflection_values'],
            'secondary_points': ret.get('secondary_points', None),
            'secondary_mask': ret.get('secondary_mask', None),
            'secondary_dir': ret.get('secondary_dir', None),
        }
        if ctx['multi'] is not None:
            B, S, R = ctx['multi']
            for key in ['idr_rgb_values', 'sg_rgb_values', 'network_object_mask', 'object_mask',
                        'sg_diffuse_rgb_values', 'sg_diffuse_albedo_values', 'sg_specular_rgb_values', 'sdf_output',
                        'points', 'sg_roughness_values', 'sg_specular_reflection_values']:
                output[key] = self.mean_pixel(output[key], B * S, R)
            output['normal_values'] = self.mean_pixel(output['normal_values'], B * S, R, vector=True)
        return output

    # ---- forward_with_point (:503-527) ------------------------------------------------------------------
    def forward_with_point(self, input):
        points = input['points']
        ray_dirs = input['ray_dirs']
        N, R, _ = points.shape
        state = self.state_freeze_geo
        self.state_freeze_geo = True
        ret = self.get_rbg_value(points.reshape(-1, 3), -ray_dirs.reshape(-1, 3))
        self.state_freeze_geo = state
        return {'idr_rgb_values': self.mean_pixel(ret['idr_rgb'], N, R),
                'sg_rgb_values': self.mean_pixel(ret['sg_rgb'], N, R)}

    # ---- get_rbg_value (:529-599) ------------------------------------------------------------------
    def get_rbg_value(self, points, view_dirs, multi_ray_data_shape=None, surface=None, unit=None):
        """unit (optional): (normals, view_dirs) already normalised (shade_tail's one-launch preparation of the hit rays)."""
        with torch.no_grad():
            # one fused pass replaces the reference's three SDF evaluations of the same points (:354, :533, :537)
            if surface is None:
                surface = self.implicit_network.value_feature_gradient(points)
            _, feature_vectors, g = surface
            if unit is not None:
                normals, view_dirs = unit
            else:
                normals = g / (torch.norm(g, dim=-1, keepdim=True) + 1e-6)
                view_dirs = view_dirs / (torch.norm(view_dirs, dim=-1, keepdim=True) + 1e-6)
        ret = {'normals': normals}
        rn = self.rendering_network
        side = None
        if rn.outputs_detached and points.is_cuda and torch.is_grad_enabled() and os.environ.get('NEFII_RADIANCE_SIDE', '0') == '1':
            # MEASURED AND NOT KEPT (round 5; NEFII_RADIANCE_SIDE=1 for A/B runs).  A radiance colour that nothing
            # differentiates (physg.conf weights it with 0) is a constant of the step, so its forward could run on a side
            # stream beside the material network's instead of ahead of it in the tail's serial chain (config 1: 67 us of an
            # 820-us chain) - and the chain did shrink to 514 us, but the step got SLOWER: config 1 0.89 -> 0.98 ms, config 2
            # 2.76 -> 3.31 (profiles/r05/tail_fusion_ab.txt).  A fifth stream (the graph's parallel branch) lands on a hardware
            # queue that one of the trace streams already uses, and the tail then waits behind a trace (round 4's
            # aliased-queues finding again).
            cur = torch.cuda.current_stream()
            side = ops.side_stream(points.device)
            side.wait_stream(cur)
            with torch.cuda.stream(side):
                idr_rgb = rn(points, normals, view_dirs, feature_vectors)
        else:
            idr_rgb = rn(points, normals, view_dirs, feature_vectors)
        mat = self.envmap_material_network(points, feature_vectors, normals)
        ret['idr_rgb'] = idr_rgb
        if self.render_type in ('pt_render_indirect_mlp', 'pt_render_indirect_mlp_memsave'):
            sg_ret = self.rgb_render(lgtSGs=mat['sg_lgtSGs'], specular_reflectance=mat['sg_specular_reflectance'],
                                     roughness=mat['sg_roughness'], diffuse_albedo=mat['sg_diffuse_albedo'],
                                     normal=normals, viewdirs=view_dirs, blending_weights=mat['sg_blending_weights'],
                                     points=points, model=self)
        else:
            sg_ret = self.rgb_render(lgtSGs=mat['sg_lgtSGs'], specular_reflectance=mat['sg_specular_reflectance'],
                                     roughness=mat['sg_roughness'], diffuse_albedo=mat['sg_diffuse_albedo'],
                                     normal=normals, viewdirs=view_dirs, blending_weights=mat['sg_blending_weights'])
        ret.update(sg_ret)
        ret.update({'sg_roughness': mat['sg_roughness'], 'sg_specular_reflectance': mat['sg_specular_reflectance'],
                    'sg_blending_weights': mat['sg_blending_weights']})
        if side is not None:
            cur.wait_stream(side)
            if not torch.cuda.is_current_stream_capturing():     # (a capture's pool outlives the graph: nothing to record)
                idr_rgb.record_stream(cur)
        return ret

    def get_background_rgb(self, light_dir):
        """sum of the light SGs along miss rays (:646-663; lobe axes normalised with +1e-8 there)."""
        if self.envmap_material_network.light_type != 'sg':
            raise NotImplementedError('2-D envmap light')
        lgt = self.envmap_material_network.get_lgtSGs()
        shape = light_dir.shape[:-1]
        return ops.EnvRadianceFn.apply(lgt, light_dir.reshape(-1, 3), 1e-8).reshape(*shape, 3)

    def mean_pixel(self, x, bs, r, vector=False):
        assert x.shape[0] == bs * r
        no_dim = x.dim() == 1
        if no_dim:
            x = x[..., None]
        x = x.reshape(bs, r, x.shape[-1])
        if vector:
            x = x[:, 0, :]
        elif x.dtype == torch.float:
            x = x.mean(1)
        elif x.dtype == torch.bool:
            x = x.all(1)
        else:
            raise TypeError('mean_pixel: undefined type %s' % x.dtype)
        if no_dim:
            x = x[..., 0]
        return x

    def get_rgb_render(self, render_type: str):
        """String -> shading function registry (:721-759).  Only the variants a shipped conf selects exist."""
        if render_type == 'sg':
            return render_with_sg
        if render_type == 'pt_render_indirect_mlp':
            from .path_tracing_render import pt_render_indirect_mlp
            return pt_render_indirect_mlp
        if render_type == 'pt_render_indirect_mlp_memsave':
            from .path_tracing_render import pt_render_indirect_mlp_memsave
            return pt_render_indirect_mlp_memsave
        raise NotImplementedError('render_type %r is not selected by any shipped conf (SURVEY.md section 2 row 5)'
                                  % render_type)
