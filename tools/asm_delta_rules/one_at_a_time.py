_N = 21
for _i in range(_N):
    RULES['only packed instruction %d as two scalar ones' % _i] = (lambda b, _i=_i: depack_some(b, lambda i, l: i == _i))
for _i in range(_N):
    RULES['all but packed instruction %d as scalar ones' % _i] = (lambda b, _i=_i: depack_some(b, lambda i, l: i != _i))
RULES['packed kept; sin / cos -> v_mul / v_fma (no transcendentals)'] = lambda b: [
    re.sub(r'v_cos_f32_e32 (v\d+), (v\d+)', r'v_fma_f32 \1, \2, 0.5, 0.5', re.sub(r'v_sin_f32_e32 (v\d+), (v\d+)', r'v_mul_f32_e32 \1, 0.5, \2', l))
    for l in b]
