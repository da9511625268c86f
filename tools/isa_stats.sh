#!/bin/bash
# register / spill / instruction-mix summary of one kernel of nefii_tracer.hip (substring of the mangled name)
K=${1:-eval_kernel16wE}
cd /root/repo/nefii_amd/csrc
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -ffp-contract=off $NEFII_EXTRA_HIPCC_FLAGS -S --cuda-device-only -o /tmp/tracer.s nefii_tracer.hip 2>/dev/null
grep -A12 "name:.*$K" /tmp/tracer.s | grep -E "vgpr|spill"
python3 - "$K" <<'PY'
import sys, re
k = sys.argv[1]
on, lines = False, []
for ln in open('/tmp/tracer.s'):
    if not on and re.match(r'^_ZN.*' + re.escape(k) + r'.*:', ln):
        on = True
    if on:
        lines.append(ln)
        if 's_endpgm' in ln:
            break
open('/tmp/k.s', 'w').writelines(lines)
txt = ''.join(lines)
for i in ['v_mfma', 'ds_write_b64', 'ds_write_b16', 'ds_read_b128', 'global_load_dwordx4', 'v_exp_f32', 'v_log_f32',
          's_waitcnt', 's_barrier', 'scratch_', 'v_cvt_pk_f16', 's_nop']:
    print(i, txt.count(i))
print('lines', len(lines))
PY
