"""`dct.dct` of the reference as its eval code uses it (train.py:57-60, generate_audio.py:22-25: `from dct.dct import IDCT`,
`IMDCT2(..., idct_op=IDCT())`).  The reference's module wraps a compiled DREAMPlace extension (dct/src/*.cpp, *.cu: absent
dependencies, not buildable -- SURVEY 8c); its `DCT` / `IDCT` compute, for either `algorithm`, exactly what
`DCT_2N_native` / `IDCT_2N_native` compute (dct/dct.py:15-17,60-64 against dct/dct_native.py:7-68: same sums, "scaled by 2 to
match other python implementation"), which this build fuses into csrc/dct.hip.  So the two names are the HIP operators under
the reference's constructor signature (expk, algorithm), and IMDCT2 / MDCT2 accept them as `idct_op` / `dct_op`.
"""
from .dct_native import DCT_2N_native, IDCT_2N_native


class DCT(DCT_2N_native):
    def __init__(self, expk=None, algorithm='N'):
        super().__init__(expk)
        if algorithm not in ('N', '2N'):
            raise ValueError("algorithm must be 'N' or '2N'")
        self.algorithm = algorithm          # both name the same transform; the HIP kernel is the N-point FFT form


class IDCT(IDCT_2N_native):
    def __init__(self, expk=None, algorithm='N'):
        super().__init__(expk)
        if algorithm not in ('N', '2N'):
            raise ValueError("algorithm must be 'N' or '2N'")
        self.algorithm = algorithm
