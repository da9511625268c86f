# base: the kernel with every packed instruction already replaced by two scalar ones (it still fails beside the evaluator)
_D = lambda b: depack_some(b, lambda i, l: True)


def _sub(body, pattern, repl):
    return [re.sub(pattern, repl, l) for l in body]


RULES = {
    'as compiled': lambda b: b,
    'no packed instructions': _D,
    'no packed, no transcendentals': lambda b: _sub(_sub(_D(b), r'v_sin_f32_e32 (v\d+), (v\d+)', r'v_mul_f32_e32 \1, 0.5, \2'),
                                                                                     r'v_cos_f32_e32 (v\d+), (v\d+)', r'v_fma_f32 \1, \2, 0.5, 0.5'),
    'no packed; SGPR operands of the loop from VGPRs': lambda b: _sub(_sub(insert(_D(b), r'^\.LBB26_13:', 'placeholder', 'before'), r'^\tplaceholder', '\tv_mov_b32_e32 v62, s0'),
                                                                      r'(v_mul_f32_e64 v6[01], v2[23]), s0', r'\1, v62'),
    'no packed; s_nop 7 after every transcendental': lambda b: insert(_D(b), TRANS, 's_nop 7\n\ts_nop 7', 'after'),
}
