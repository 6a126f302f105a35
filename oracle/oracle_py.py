"""TEST INFRASTRUCTURE ONLY -- ctypes binding of oracle/liboracle.so.

Importable from tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg, as the
checker only.  Nothing under tfhe_amd/ imports this module.
"""
import ctypes as C
import pathlib
import subprocess

import numpy as np

HERE = pathlib.Path(__file__).resolve().parent
LIB_PATH = HERE / "liboracle.so"

SCHOOLBOOK, NTT = 0, 1


class Params(C.Structure):
    _fields_ = [("n", C.c_int32), ("N", C.c_int32), ("K", C.c_int32), ("bg_bit", C.c_int32),
                ("L", C.c_int32), ("T", C.c_int32), ("base_bit", C.c_int32),
                ("alpha_lv0", C.c_double), ("alpha_lv1", C.c_double)]

    @property
    def n1(self):
        return self.n + 1

    @property
    def lv1(self):
        return self.N * self.K

    @property
    def rows(self):
        return (self.K + 1) * self.L

    def bk_shape(self):
        return (self.n, self.rows, self.K + 1, self.N)

    def ks_shape(self):
        return (self.lv1, self.T, 1 << self.base_bit, self.n + 1)


class Rng(C.Structure):
    _fields_ = [("s", C.c_uint64 * 4), ("has_spare", C.c_int), ("spare", C.c_double)]


def build():
    subprocess.run(["make", "-C", str(HERE), "liboracle.so"], check=True, capture_output=True)


_lib = None


def lib():
    global _lib
    if _lib is None:
        if not LIB_PATH.exists():
            build()
        _lib = C.CDLL(str(LIB_PATH))
        _lib.tfo_ctx_create.restype = C.c_void_p
        _lib.tfo_bk_words.restype = C.c_size_t
        _lib.tfo_ks_words.restype = C.c_size_t
        _lib.tfo_tlwe_phase.restype = C.c_uint32
        _lib.tfo_torus_from_double.restype = C.c_uint32
        _lib.tfo_torus_from_double.argtypes = [C.c_double]
        _lib.tfo_rng_u64.restype = C.c_uint64
        _lib.tfo_rng_normal.restype = C.c_double
    return _lib


def _p(a):
    return a.ctypes.data_as(C.c_void_p)


def _u32(a):
    return np.ascontiguousarray(a, dtype=np.uint32)


def params_default():
    p = Params()
    lib().tfo_params_default(C.byref(p))
    return p


def params_n2048():
    p = Params()
    lib().tfo_params_n2048(C.byref(p))
    return p


def params_from(n, N, K, bg_bit, L, T, base_bit, alpha_lv0=9.5367431640625e-07, alpha_lv1=2.9802322387695312e-08):
    return Params(int(n), int(N), int(K), int(bg_bit), int(L), int(T), int(base_bit), alpha_lv0, alpha_lv1)


def rng(seed):
    r = Rng()
    lib().tfo_rng_seed(C.byref(r), C.c_uint64(seed))
    return r


# ---- primitives -------------------------------------------------------------
def poly_mul_x_exp(poly, shift):
    poly = _u32(poly)
    out = np.empty_like(poly)
    lib().tfo_poly_mul_x_exp(_p(out), _p(poly), C.c_int(poly.shape[-1]), C.c_int(int(shift)))
    return out


def decompose(poly, p):
    poly = _u32(poly)
    out = np.empty((p.L, p.N), dtype=np.int32)
    lib().tfo_decompose(_p(out), _p(poly), C.byref(p))
    return out


def negacyclic_mul(d, b, mode):
    d = np.ascontiguousarray(d, dtype=np.int32)
    b = _u32(b)
    out = np.empty_like(b)
    lib().tfo_negacyclic_mul(_p(out), _p(d), _p(b), C.c_int(b.shape[-1]), C.c_int(mode))
    return out


def external_product(trgsw, trlwe, p, mode):
    trgsw, trlwe = _u32(trgsw), _u32(trlwe)
    out = np.empty((p.K + 1, p.N), dtype=np.uint32)
    lib().tfo_external_product(_p(out), _p(trgsw), _p(trlwe), C.byref(p), C.c_int(mode))
    return out


def cmux(trgsw, d0, d1, p, mode):
    trgsw, d0, d1 = _u32(trgsw), _u32(d0), _u32(d1)
    out = np.empty((p.K + 1, p.N), dtype=np.uint32)
    lib().tfo_cmux(_p(out), _p(trgsw), _p(d0), _p(d1), C.byref(p), C.c_int(mode))
    return out


def test_vector(p):
    tv = np.zeros((p.K + 1, p.N), dtype=np.uint32)
    tv[0, :] = 1 << 29
    return tv


def blind_rotate(c, bk, p, mode, tv=None, trace=False):
    c, bk = _u32(c), _u32(bk)
    tv = test_vector(p) if tv is None else _u32(tv)
    acc = np.empty((p.K + 1, p.N), dtype=np.uint32)
    steps = []
    FN = C.CFUNCTYPE(None, C.c_void_p, C.c_int, C.POINTER(C.c_uint32))

    def cb(_user, _step, ptr):
        steps.append(np.ctypeslib.as_array(ptr, shape=(p.K + 1, p.N)).copy())

    fn = FN(cb) if trace else C.cast(None, FN)
    lib().tfo_blind_rotate(_p(acc), _p(c), _p(bk), _p(tv), C.byref(p), C.c_int(mode), fn, None)
    return (acc, np.stack(steps)) if trace else acc


def sample_extract_index(trlwe, x, p):
    trlwe = _u32(trlwe)
    out = np.empty(p.lv1 + 1, dtype=np.uint32)
    lib().tfo_sample_extract_index(_p(out), _p(trlwe), C.c_int(int(x)), C.byref(p))
    return out


def gate_bootstrapping(c, bk, p, mode):
    c, bk = _u32(c), _u32(bk)
    out = np.empty(p.lv1 + 1, dtype=np.uint32)
    lib().tfo_gate_bootstrapping_tlwe_to_tlwe(_p(out), _p(c), _p(bk), C.byref(p), C.c_int(mode))
    return out


def identity_key_switch(lv1, ks, p):
    lv1, ks = _u32(lv1), _u32(ks)
    out = np.empty(p.n + 1, dtype=np.uint32)
    lib().tfo_identity_key_switch(_p(out), _p(lv1), _p(ks), C.byref(p))
    return out


def nand_combine(a, b, p):
    a, b = _u32(a), _u32(b)
    out = np.empty(p.n + 1, dtype=np.uint32)
    lib().tfo_nand_combine(_p(out), _p(a), _p(b), C.byref(p))
    return out


GATES = {"nand": 0, "and": 1, "or": 2, "nor": 3, "xor": 4, "xnor": 5, "andny": 6, "andyn": 7, "orny": 8, "oryn": 9}
GATE_TRUTH = {"nand": lambda a, b: 1 - (a & b), "and": lambda a, b: a & b, "or": lambda a, b: a | b,
              "nor": lambda a, b: 1 - (a | b), "xor": lambda a, b: a ^ b, "xnor": lambda a, b: 1 - (a ^ b),
              "andny": lambda a, b: (1 - a) & b, "andyn": lambda a, b: a & (1 - b),
              "orny": lambda a, b: (1 - a) | b, "oryn": lambda a, b: a | (1 - b)}


def gate_combine(op, a, b, p):
    a, b = _u32(a), _u32(b)
    out = np.empty(p.n + 1, dtype=np.uint32)
    rc = lib().tfo_gate_combine(_p(out), C.c_int(GATES[op] if isinstance(op, str) else int(op)), _p(a), _p(b), C.byref(p))
    if rc:
        raise ValueError(f"unknown gate {op}")
    return out


def nand(a, b, bk, ks, p, mode):
    a, b, bk, ks = _u32(a), _u32(b), _u32(bk), _u32(ks)
    out = np.empty(p.n + 1, dtype=np.uint32)
    lib().tfo_nand(_p(out), _p(a), _p(b), _p(bk), _p(ks), C.byref(p), C.c_int(mode))
    return out


# ---- fast context -----------------------------------------------------------
class Context:
    def __init__(self, p, bk, ks):
        self.p = p
        self.bk, self.ks = _u32(bk), _u32(ks)  # keep alive: ks is borrowed by the C side
        assert self.bk.size == lib().tfo_bk_words(C.byref(p)) and self.ks.size == lib().tfo_ks_words(C.byref(p))
        self.h = C.c_void_p(lib().tfo_ctx_create(C.byref(p), _p(self.bk), _p(self.ks)))
        assert self.h.value, "invalid parameters"

    def close(self):
        if self.h:
            lib().tfo_ctx_destroy(self.h)
            self.h = None

    def __del__(self):
        self.close()

    def blind_rotate(self, c):
        c = _u32(c)
        acc = np.empty((self.p.K + 1, self.p.N), dtype=np.uint32)
        lib().tfo_ctx_blind_rotate(self.h, _p(acc), _p(c))
        return acc

    def nand_batch(self, a, b, threads=0):
        a, b = _u32(a), _u32(b)
        out = np.empty_like(a)
        lib().tfo_ctx_nand_batch(self.h, _p(out), _p(a), _p(b), C.c_size_t(a.shape[0]), C.c_int(threads))
        return out

    def gate_batch(self, op, a, b, threads=0):
        """op: a gate name / code for all gates, or a uint8 array of codes, one per gate."""
        a, b = _u32(a), _u32(b)
        out = np.empty_like(a)
        ops = None
        code = 0
        if isinstance(op, np.ndarray):
            ops = np.ascontiguousarray(op, dtype=np.uint8)
            assert ops.shape[0] == a.shape[0]
        else:
            code = GATES[op] if isinstance(op, str) else int(op)
        rc = lib().tfo_ctx_gate_batch(self.h, C.c_int(code), _p(ops) if ops is not None else None, _p(out), _p(a),
                                      _p(b), C.c_size_t(a.shape[0]), C.c_int(threads))
        if rc:
            raise ValueError("unknown gate code")
        return out

    def mux_batch(self, sel, x1, x0, threads=0):
        sel, x1, x0 = _u32(sel), _u32(x1), _u32(x0)
        out = np.empty_like(sel)
        lib().tfo_ctx_mux_batch(self.h, _p(out), _p(sel), _p(x1), _p(x0), C.c_size_t(sel.shape[0]), C.c_int(threads))
        return out

    def bootstrap_batch(self, c, threads=0):
        c = _u32(c)
        out = np.empty_like(c)
        lib().tfo_ctx_bootstrap_batch(self.h, _p(out), _p(c), C.c_size_t(c.shape[0]), C.c_int(threads))
        return out


def max_threads():
    return lib().tfo_max_threads()


# ---- seeded client side -----------------------------------------------------
def keygen_secret(p, r):
    s0 = np.empty(p.n, dtype=np.uint8)
    sr = np.empty((p.K, p.N), dtype=np.uint8)
    lib().tfo_keygen_secret(_p(s0), _p(sr), C.byref(p), C.byref(r))
    return s0, sr


def make_bk(s0, sr, p, r):
    bk = np.empty(p.bk_shape(), dtype=np.uint32)
    lib().tfo_make_bk(_p(bk), _p(s0), _p(sr), C.byref(p), C.byref(r))
    return bk


def make_ks(s0, sr, p, r):
    ks = np.empty(p.ks_shape(), dtype=np.uint32)
    lib().tfo_make_ks(_p(ks), _p(s0), _p(sr), C.byref(p), C.byref(r))
    return ks


def chacha20_block(seed_words, stream, row, index):
    out = np.empty(16, dtype=np.uint32)
    sw = np.ascontiguousarray(seed_words, dtype=np.uint32)
    lib().tfo_chacha20_block(_p(out), _p(sw), C.c_uint32(stream), C.c_uint64(row), C.c_uint32(index))
    return out


def keygen_chacha(s0, sr, p, alpha_lv0, alpha_lv1, seed_words):
    """(bk, ks) as the GPU key generator makes them from the same secret key and 256-bit seed."""
    sw = np.ascontiguousarray(seed_words, dtype=np.uint32)
    s0 = np.ascontiguousarray(s0, dtype=np.uint8)
    sr = np.ascontiguousarray(sr, dtype=np.uint8)
    bk = np.empty(p.bk_shape(), dtype=np.uint32)
    ks = np.empty(p.ks_shape(), dtype=np.uint32)
    lib().tfo_keygen_chacha_bk(_p(bk), _p(s0), _p(sr), C.byref(p), C.c_double(alpha_lv1), _p(sw))
    lib().tfo_keygen_chacha_ks(_p(ks), _p(s0), _p(sr), C.byref(p), C.c_double(alpha_lv0), _p(sw))
    return bk, ks


def encrypt_bits(bits, s0, p, r):
    bits = np.asarray(bits).astype(np.int32).ravel()
    out = np.empty((bits.size, p.n + 1), dtype=np.uint32)
    for g, bit in enumerate(bits):
        lib().tfo_encrypt_bit(_p(out[g]), C.c_int(int(bit)), _p(s0), C.byref(p), C.byref(r))
    return out


def tlwe_encrypt(m, s, alpha, r):
    ct = np.empty(s.size + 1, dtype=np.uint32)
    lib().tfo_tlwe_encrypt(_p(ct), C.c_uint32(int(m)), _p(s), C.c_int(s.size), C.c_double(alpha), C.byref(r))
    return ct


def tlwe_phase(ct, s):
    ct = _u32(ct)
    return int(lib().tfo_tlwe_phase(_p(ct), _p(s), C.c_int(s.size)))


def decrypt_bits(cts, s0, p):
    cts = _u32(cts).reshape(-1, p.n + 1)
    return np.array([lib().tfo_decrypt_bit(_p(ct), _p(s0), C.byref(p)) for ct in cts], dtype=np.uint8)


def phases(cts, s0, p):
    cts = _u32(cts).reshape(-1, p.n + 1)
    return np.array([tlwe_phase(ct, s0) for ct in cts], dtype=np.uint32)


def trlwe_encrypt(m, sr, p, r):
    m = _u32(m)
    ct = np.empty((p.K + 1, p.N), dtype=np.uint32)
    lib().tfo_trlwe_encrypt(_p(ct), _p(m), _p(sr), C.byref(p), C.byref(r))
    return ct


def trlwe_phase(ct, sr, p):
    ct = _u32(ct)
    m = np.empty(p.N, dtype=np.uint32)
    lib().tfo_trlwe_phase(_p(m), _p(ct), _p(sr), C.byref(p))
    return m


def trgsw_encrypt_int_poly(mu, sr, p, r):
    mu = np.ascontiguousarray(mu, dtype=np.int32)
    out = np.empty((p.rows, p.K + 1, p.N), dtype=np.uint32)
    lib().tfo_trgsw_encrypt_int_poly(_p(out), _p(mu), _p(sr), C.byref(p), C.byref(r))
    return out


def keyset(p, seed):
    """Deterministic (secret lv0, secret ring, bk, ks) for a seed."""
    r = rng(seed)
    s0, sr = keygen_secret(p, r)
    return s0, sr, make_bk(s0, sr, p, r), make_ks(s0, sr, p, r)
