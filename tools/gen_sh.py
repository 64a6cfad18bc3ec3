#!/usr/bin/env python3
"""Generate the real-spherical-harmonics polynomial tables used by
 (a) oracle/sh_table.inc          -- double precision, un-optimised, one expression per line
 (b) torch-ngp_amd/csrc/sh_poly.inc -- fp32 device code, common sub-expressions shared

The 64 basis polynomials (bands 0..7) are the Cartesian forms the reference evaluates
(reference: shencoder/src/shencoder.cu:50-120, closed forms given in its trailing comments);
they are restated here symbolically with exact radicals.  The partial derivatives are NOT
transcribed from the reference (shencoder.cu:130-352): they are obtained by symbolic
differentiation of the value polynomials, so a transcription slip in a value polynomial cannot
hide behind a matching slip in a derivative table.

The value polynomials are pinned independently in tests/test_oracle_sh.py against
scipy.special.sph_harm (all 64, unit vectors) and against golden vectors produced by the
reference's own pure-torch SHEncoder_torch (testing/test_shencoder.py:8-89, bands 0..4).

Run:  python tools/gen_sh.py       (rewrites both .inc files in place; needs sympy)
"""
import os
import sympy as sp

x, y, z = sp.symbols('x y z', real=True)
pi = sp.pi
s = sp.sqrt
x2, y2, z2 = x * x, y * y, z * z
x4, y4, z4 = x2 * x2, y2 * y2, z2 * z2
x6, y6, z6 = x4 * x2, y4 * y2, z4 * z2
xy, xz, yz = x * y, x * z, y * z
xyz = x * y * z


def basis():
    """64 real SH polynomials, index = l*l + (m + l)."""
    Y = [None] * 64
    # band 0
    Y[0] = 1 / (2 * s(pi))
    # band 1
    Y[1] = -s(3) * y / (2 * s(pi))
    Y[2] = s(3) * z / (2 * s(pi))
    Y[3] = -s(3) * x / (2 * s(pi))
    # band 2
    Y[4] = s(15) * xy / (2 * s(pi))
    Y[5] = -s(15) * yz / (2 * s(pi))
    Y[6] = s(5) * (3 * z2 - 1) / (4 * s(pi))
    Y[7] = -s(15) * xz / (2 * s(pi))
    Y[8] = s(15) * (x2 - y2) / (4 * s(pi))
    # band 3
    Y[9] = s(70) * y * (-3 * x2 + y2) / (8 * s(pi))
    Y[10] = s(105) * xyz / (2 * s(pi))
    Y[11] = s(42) * y * (1 - 5 * z2) / (8 * s(pi))
    Y[12] = s(7) * z * (5 * z2 - 3) / (4 * s(pi))
    Y[13] = s(42) * x * (1 - 5 * z2) / (8 * s(pi))
    Y[14] = s(105) * z * (x2 - y2) / (4 * s(pi))
    Y[15] = s(70) * x * (-x2 + 3 * y2) / (8 * s(pi))
    # band 4
    Y[16] = 3 * s(35) * xy * (x2 - y2) / (4 * s(pi))
    Y[17] = 3 * s(70) * yz * (-3 * x2 + y2) / (8 * s(pi))
    Y[18] = 3 * s(5) * xy * (7 * z2 - 1) / (4 * s(pi))
    Y[19] = 3 * s(10) * yz * (3 - 7 * z2) / (8 * s(pi))
    Y[20] = 3 * (-30 * z2 + 35 * z4 + 3) / (16 * s(pi))
    Y[21] = 3 * s(10) * xz * (3 - 7 * z2) / (8 * s(pi))
    Y[22] = 3 * s(5) * (x2 - y2) * (7 * z2 - 1) / (8 * s(pi))
    Y[23] = 3 * s(70) * xz * (-x2 + 3 * y2) / (8 * s(pi))
    Y[24] = 3 * s(35) * (-6 * x2 * y2 + x4 + y4) / (16 * s(pi))
    # band 5
    Y[25] = 3 * s(154) * y * (10 * x2 * y2 - 5 * x4 - y4) / (32 * s(pi))
    Y[26] = 3 * s(385) * xyz * (x2 - y2) / (4 * s(pi))
    Y[27] = -s(770) * y * (3 * x2 - y2) * (9 * z2 - 1) / (32 * s(pi))
    Y[28] = s(1155) * xyz * (3 * z2 - 1) / (4 * s(pi))
    Y[29] = s(165) * y * (14 * z2 - 21 * z4 - 1) / (16 * s(pi))
    Y[30] = s(11) * z * (-70 * z2 + 63 * z4 + 15) / (16 * s(pi))
    Y[31] = s(165) * x * (14 * z2 - 21 * z4 - 1) / (16 * s(pi))
    Y[32] = s(1155) * z * (x2 - y2) * (3 * z2 - 1) / (8 * s(pi))
    Y[33] = -s(770) * x * (x2 - 3 * y2) * (9 * z2 - 1) / (32 * s(pi))
    Y[34] = 3 * s(385) * z * (-6 * x2 * y2 + x4 + y4) / (16 * s(pi))
    Y[35] = 3 * s(154) * x * (10 * x2 * y2 - x4 - 5 * y4) / (32 * s(pi))
    # band 6
    Y[36] = s(6006) * xy * (-10 * x2 * y2 + 3 * x4 + 3 * y4) / (32 * s(pi))
    Y[37] = 3 * s(2002) * yz * (10 * x2 * y2 - 5 * x4 - y4) / (32 * s(pi))
    Y[38] = 3 * s(91) * xy * (x2 - y2) * (11 * z2 - 1) / (8 * s(pi))
    Y[39] = -s(2730) * yz * (3 * x2 - y2) * (11 * z2 - 3) / (32 * s(pi))
    Y[40] = s(2730) * xy * (-18 * z2 + 33 * z4 + 1) / (32 * s(pi))
    Y[41] = s(273) * yz * (30 * z2 - 33 * z4 - 5) / (16 * s(pi))
    Y[42] = s(13) * (105 * z2 - 315 * z4 + 231 * z6 - 5) / (32 * s(pi))
    Y[43] = s(273) * xz * (30 * z2 - 33 * z4 - 5) / (16 * s(pi))
    Y[44] = s(2730) * (x2 - y2) * (11 * z2 * (3 * z2 - 1) - 7 * z2 + 1) / (64 * s(pi))
    Y[45] = -s(2730) * xz * (x2 - 3 * y2) * (11 * z2 - 3) / (32 * s(pi))
    Y[46] = 3 * s(91) * (11 * z2 - 1) * (-6 * x2 * y2 + x4 + y4) / (32 * s(pi))
    Y[47] = 3 * s(2002) * xz * (10 * x2 * y2 - x4 - 5 * y4) / (32 * s(pi))
    Y[48] = s(6006) * (15 * x2 * y4 - 15 * x4 * y2 + x6 - y6) / (64 * s(pi))
    # band 7
    Y[49] = 3 * s(715) * y * (-21 * x2 * y4 + 35 * x4 * y2 - 7 * x6 + y6) / (64 * s(pi))
    Y[50] = 3 * s(10010) * xyz * (-10 * x2 * y2 + 3 * x4 + 3 * y4) / (32 * s(pi))
    Y[51] = -3 * s(385) * y * (13 * z2 - 1) * (-10 * x2 * y2 + 5 * x4 + y4) / (64 * s(pi))
    Y[52] = 3 * s(385) * xyz * (x2 - y2) * (13 * z2 - 3) / (8 * s(pi))
    Y[53] = -3 * s(35) * y * (3 * x2 - y2) * (13 * z2 * (11 * z2 - 3) - 27 * z2 + 3) / (64 * s(pi))
    Y[54] = 3 * s(70) * xyz * (-110 * z2 + 143 * z4 + 15) / (32 * s(pi))
    Y[55] = s(105) * y * (-135 * z2 + 495 * z4 - 429 * z6 + 5) / (64 * s(pi))
    Y[56] = s(15) * z * (315 * z2 - 693 * z4 + 429 * z6 - 35) / (32 * s(pi))
    Y[57] = s(105) * x * (-135 * z2 + 495 * z4 - 429 * z6 + 5) / (64 * s(pi))
    Y[58] = s(70) * z * (x2 - y2) * (143 * z2 * (3 * z2 - 1) - 187 * z2 + 45) / (64 * s(pi))
    Y[59] = -3 * s(35) * x * (x2 - 3 * y2) * (13 * z2 * (11 * z2 - 3) - 27 * z2 + 3) / (64 * s(pi))
    Y[60] = 3 * s(385) * z * (13 * z2 - 3) * (-6 * x2 * y2 + x4 + y4) / (32 * s(pi))
    Y[61] = -3 * s(385) * x * (13 * z2 - 1) * (-10 * x2 * y2 + x4 + 5 * y4) / (64 * s(pi))
    Y[62] = 3 * s(10010) * z * (15 * x2 * y4 - 15 * x4 * y2 + x6 - y6) / (64 * s(pi))
    Y[63] = 3 * s(715) * x * (-35 * x2 * y4 + 21 * x4 * y2 - x6 + 7 * y6) / (64 * s(pi))
    return Y


def c_expr(e, suffix):
    """C expression for a sympy polynomial with all numeric factors evaluated to 17 digits."""
    e = sp.nsimplify(e)
    e = sp.expand(e)
    # Horner in z then y then x keeps the operation count and rounding modest
    e = sp.horner(e, z, y, x) if e.free_symbols else e
    e = e.evalf(17)
    code = sp.ccode(e)
    if suffix:
        # turn every floating literal into a float literal
        import re
        code = re.sub(r'(?<![\w.])(\d+\.\d*(?:[eE][-+]?\d+)?|\d+[eE][-+]?\d+)(?![\w.])', r'\1f', code)
        code = re.sub(r'(?<![\w.])(\d+)(?![\w.\[])', r'\1.0f', code)
        code = code.replace('pow(', 'powf(')
    return code


def expand_pows(code):
    """sympy emits pow(x, n); rewrite small integer powers as products (exact same rounding
    order on host and device, no libm call)."""
    import re
    pat = re.compile(r'powf?\(([xyz]), (\d+)(?:\.0f)?\)')

    def rep(m):
        v, n = m.group(1), int(m.group(2))
        return '(' + '*'.join([v] * n) + ')'
    return pat.sub(rep, code)


def main():
    here = os.path.dirname(os.path.abspath(__file__))
    root = os.path.dirname(here)
    Y = basis()
    dY = [[sp.diff(Yi, v) for Yi in Y] for v in (x, y, z)]

    # ---------------- oracle (double) ----------------
    lines = ['// GENERATED by tools/gen_sh.py -- do not edit.  Double-precision restatement of the',
             '// polynomials evaluated by the reference kernel (shencoder/src/shencoder.cu:50-120) and',
             '// their symbolic partial derivatives.  Included by oracle/ngp_oracle.c only.',
             '#define ORC_SH_MAX 64',
             'static void orc_sh_eval(double x, double y, double z, double *Y, double *dYx, double *dYy, double *dYz) {']
    for i in range(64):
        lines.append('    Y[%d] = %s;' % (i, expand_pows(c_expr(Y[i], False))))
    lines.append('    if (!dYx) return;')
    for name, tab in (('dYx', dY[0]), ('dYy', dY[1]), ('dYz', dY[2])):
        for i in range(64):
            lines.append('    %s[%d] = %s;' % (name, i, expand_pows(c_expr(tab[i], False))))
    lines.append('}')
    with open(os.path.join(root, 'oracle', 'sh_table.inc'), 'w') as f:
        f.write('\n'.join(lines) + '\n')

    # ---------------- device (float), per band so that `if (bands <= k) return` works ----------------
    dl = ['// GENERATED by tools/gen_sh.py -- do not edit.  fp32 device evaluation of the real SH',
          '// basis, band by band (band b contributes components b*b .. (b+1)*(b+1)-1).',
          '// SH_OUT(i, v) / SH_DX(i, v) / SH_DY(i, v) / SH_DZ(i, v) are supplied by the includer.']
    for band in range(8):
        lo, hi = band * band, (band + 1) * (band + 1)
        dl.append('#define SH_BAND_%d_VALUES \\' % band)
        for i in range(lo, hi):
            dl.append('    SH_OUT(%d, %s); \\' % (i, expand_pows(c_expr(Y[i], True))))
        dl.append('    ((void)0)')
        dl.append('#define SH_BAND_%d_GRADS \\' % band)
        for i in range(lo, hi):
            dl.append('    SH_DX(%d, %s); \\' % (i, expand_pows(c_expr(dY[0][i], True))))
            dl.append('    SH_DY(%d, %s); \\' % (i, expand_pows(c_expr(dY[1][i], True))))
            dl.append('    SH_DZ(%d, %s); \\' % (i, expand_pows(c_expr(dY[2][i], True))))
        dl.append('    ((void)0)')
    with open(os.path.join(root, 'torch-ngp_amd', 'csrc', 'sh_poly.inc'), 'w') as f:
        f.write('\n'.join(dl) + '\n')
    print('wrote oracle/sh_table.inc and torch-ngp_amd/csrc/sh_poly.inc')


if __name__ == '__main__':
    main()
