"""bench.py's LAST stdout line is what the driver parses: a flat JSON object below 4 KB (round 5's single 25.8 KB line was not
parsed).  compact_line is a pure function of the full record, so it is held to that here on canned records -
round 5's own full record (profiles/r05/bench_default.json, the one that did not parse) and a synthetic worst case.
Also: the parity suite's soft mode cannot be switched on by an environment variable any more."""
import json
import os
import subprocess
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)

REQUIRED = ('metric', 'value', 'unit', 'n_gpus', 'steps', 'warmup', 'ms_per_step', 'higher_is_better', 'scaling', 'vs_baseline',
            'dtype', 'data', 'config', 'roofline', 'cpu_baseline')


def _canned():
    """a full record of today's shape with every optional block present and over-long strings everywhere"""
    blob = 'x' * 3000
    roof = {'bound': 'mfma', 'kernel': 'eval_kernel16q + eval_kernel16s (fused SDF MLP over the tracer work list)', 'achieved': 698.123456,
            'peak': 2500.0, 'unit': 'TFLOP/s', 'frac': 0.27912345, 'frac_executed': 0.27912345, 'frac_credited': 0.5059, 'frac_kernel': 0.5059,
            'frac_step': 0.2707, 'frac_chip': 0.4, 'frac_8d': 0.325, 'issued_tflops': 1071.2, 'issued_frac_of_sustained': 0.6259,
            'sustained_peak': {'value': 1712.3, 'what': blob}, 'board_power': {'avg_w': 1251.0, 'cap_w': 1400.0, 'source': blob},
            'frac_definitions': blob, 'arithmetic': blob, 'traffic': 298.3e6, 'traffic_source': blob, 'traffic_algorithmic': 20.1e6,
            'flops_per_sdf_eval': 3671040, 'sdf_evals_per_step': 38686375, 'sdf_evals_executed_split_precision': 5712684,
            'sdf_evals_executed_single_pass': 15634010, 'coarse_tau': 0.001168, 'coarse_audit_max': 4.5e-4,
            'coarse_audit_events': [['recalibrated', 1e-3, 2e-3]] * 50, 'minsdf_lipschitz': 2.074, 'minsdf_lipschitz_violation': 0.0,
            'lipschitz_audited_samples': 123456, 'kernel_ms_per_step': 112.3, 'launches_per_step': 78, 'hit_fraction': 0.4441,
            'secondary_hit_fraction': 0.5239, 'step_model': {'formula': blob, 'A_flops_per_primary_ray': 3.69e8}}
    cfg = {'workload': 'cfg3: ' + blob, 'primary_rays_per_step_per_gpu': 262144, 'parallelism': 'dp1', 'trace_tier': True,
           'secondary_miss_search': False, 'step_graph': False, 'cycle_batches': 4, 'trace_prefetch': 3, 'nonfinite_steps': 0, 'loss': 0.72}
    nested = {'value': 1.0e6, 'ms_per_step': 2.4, 'config': dict(cfg), 'roofline': dict(roof)}
    return {'metric': 'training rays/sec (Step-2 material opt)', 'value': 1832850.123, 'unit': 'rays/s', 'n_gpus': 1, 'steps': 20,
            'warmup': 3, 'ms_per_step': 143.0312345, 'ms_per_step_repeats': [143.1, 142.2, 143.0], 'higher_is_better': True, 'scaling': 'weak',
            'vs_baseline': None, 'dtype': 'f16x3', 'data': 'synthetic', 'config': cfg, 'roofline': roof,
            'cfg1': dict(nested), 'cfg2': dict(nested), 'cfg4': dict(nested), 'cfg5': {'ms_per_step': 4212.0, 'config': {'workload': blob}},
            'cfg3_replicated_stand_in': {'ms_per_step': 159.8, 'note': blob}, 'cfg3_zero_padded_stand_in': {'ms_per_step': 131.0, 'note': blob},
            'untiered': {'ms_per_step': 169.1},
            'cpu_baseline': {'value': 592.5, 'unit': 'rays/s', 'cores': 16, 'kind': 'port', 'value_1_thread': 122.2,
                             'host_cpu': 'AMD EPYC 9575F 64-Core Processor', 'host_cores': 128, 'sample': blob},
            'parity_vs_cpu_oracle': {'rgb_rel_l2': 1.378e-4, 'albedo_rel_l2': 9.357e-5, 'rgb_rel_l2_same_samples': 3.3e-5,
                                     'tolerance_rel_l2': 1e-3, 'hit_pixels': 60, 'pixels': 128, 'hit_mask_mismatches': 0,
                                     'rays_with_another_sampled_direction': 23, 'trace_tier': True, 'sample': blob},
            'ms_per_step_without_dead_min_sdf_search': 115.68, 'invalid': False}


def _check(line, full):
    assert '\n' not in line
    assert len(line) < 4096, len(line)
    back = json.loads(line)
    assert json.loads(json.dumps(back)) == back                 # round-trips
    for k in REQUIRED:
        assert k in back, k
    assert back['value'] == pytest.approx(full['value'], rel=1e-5)
    assert back['ms_per_step'] == pytest.approx(full['ms_per_step'], rel=1e-4)
    assert isinstance(back['config']['workload'], str) and len(back['config']['workload']) < 32      # a name, no prose
    assert not any(k in back['config'] for k in ('model', 'global_batch', 'seq_len'))                 # no model keys
    r = back['roofline']
    for k in ('bound', 'achieved', 'peak', 'unit', 'frac', 'traffic'):
        assert k in r, k
    assert r['frac'] == pytest.approx(r['achieved'] / r['peak'], rel=2e-3)
    c = back['cpu_baseline']
    for k in ('value', 'unit', 'cores', 'kind', 'sample'):
        assert k in c, k
    # nothing nested deeper than one object, no string longer than a label
    for k, v in back.items():
        if isinstance(v, dict):
            assert not any(isinstance(x, (dict, list)) for x in v.values()), k
            assert all(len(x) <= 128 for x in v.values() if isinstance(x, str)), k
    return back


def test_compact_line_of_a_canned_full_record():
    import bench
    full = _canned()
    back = _check(bench.compact_line(full), full)
    # frac is what the kernels executed; the credited (reference-count) figure rides beside it under its own name
    assert back['roofline']['frac'] == pytest.approx(0.2791, rel=1e-3) and back['roofline']['frac_credited'] == pytest.approx(0.5059)
    assert back['config']['trace_tier'] is True and back['ms_per_step_library_defaults'] == pytest.approx(169.1)
    assert back['parity']['flips'] == 0 and back['others']['cfg4_ms_per_step'] == pytest.approx(2.4)


def test_compact_line_of_round_5s_unparsed_record():
    import bench
    path = os.path.join(ROOT, 'profiles', 'r05', 'bench_default.json')
    if not os.path.exists(path):
        pytest.skip('profiles/r05/bench_default.json not in this tree')
    full = json.load(open(path))
    assert len(json.dumps(full)) > 20000                # the line the driver could not parse
    full['roofline']['achieved'] = full['roofline']['frac_executed'] * full['roofline']['peak']      # (round 6's key meaning)
    full['roofline']['frac_credited'], full['roofline']['frac'] = full['roofline']['frac'], full['roofline']['frac_executed']
    _check(bench.compact_line(full), full)


def test_compact_line_sheds_optional_blocks_rather_than_grow():
    import bench
    full = _canned()
    full['config'].update({'k%d' % i: 1 for i in range(8)})
    full['cpu_baseline']['host_cpu'] = 'y' * 120
    line = bench.compact_line(full)
    assert len(line) < 4096 and all(k in json.loads(line) for k in REQUIRED)
    # a render record (config 5) has no roofline block of its own and still makes a line
    render = {'metric': 'render rays/sec', 'value': 2.5e6, 'unit': 'rays/s', 'n_gpus': 1, 'steps': 1, 'warmup': 0, 'ms_per_step': 63400.0,
              'higher_is_better': True, 'scaling': 'strong', 'vs_baseline': None, 'dtype': 'f16x3', 'data': 'synthetic',
              'config': {'workload': 'cfg5: ' + 'z' * 500, 'primary_rays_per_frame': 163840000, 'parallelism': 'pixel chunks x 1'},
              'invalid': False}
    back = json.loads(bench.compact_line(render))
    assert back['config']['workload'] == 'cfg5' and len(json.dumps(back)) < 1024


def test_parity_soft_mode_is_not_reachable_through_the_environment():
    sys.path.insert(0, os.path.join(ROOT, 'tests'))
    import parity
    assert parity.SOFT is False or '--parity-soft' in sys.argv
    env = dict(os.environ, NEFII_PARITY_SOFT='1')
    r = subprocess.run([sys.executable, '-m', 'pytest', os.path.join(ROOT, 'tests', 'test_conf_cpu.py'), '-q', '-x', '-m', 'not gpu',
                        '-p', 'no:cacheprovider'], env=env, capture_output=True, text=True, cwd=ROOT, timeout=300)
    assert r.returncode != 0 and 'NEFII_PARITY_SOFT' in (r.stdout + r.stderr)
