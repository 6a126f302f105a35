"""ctypes binding of the C ABI declared in include/tfhe_amd.h.

Method names follow the reference's interface for this path: HomNAND::nand
(reference src/hom_nand.hpp:74-81), TRGSW::gate_bootstrapping_tlwe_to_tlwe
(src/trgsw.hpp:172-182), TLWEKeySwitching::identity_key_switch
(src/tlwe_key_switching.hpp:47-65).  Arrays are numpy uint32 on the host or torch uint32/int32
tensors on the GPU; the binding only passes their addresses.
"""
import ctypes as C
import os
import pathlib

import numpy as np

_HERE = pathlib.Path(__file__).resolve().parent
# TFHE_AMD_LIBRARY selects another build of the same library (kernel A/B measurements)
_LIB_FILE = pathlib.Path(os.environ.get("TFHE_AMD_LIBRARY") or (_HERE / "libtfhe_amd.so"))
_lib = None
_load_error = None


class TfheAmdError(RuntimeError):
    pass


class Params(C.Structure):
    """Run-time mirror of the reference's parameter structs (src/tlwe.hpp:11-14,
    src/trlwe.hpp:9-18, src/trgsw.hpp:6-10, src/tlwe_key_switching.hpp:6-9)."""
    _fields_ = [("n", C.c_int32), ("N", C.c_int32), ("K", C.c_int32), ("bg_bit", C.c_int32),
                ("L", C.c_int32), ("T", C.c_int32), ("base_bit", C.c_int32)]

    @property
    def n1(self):
        return self.n + 1

    @property
    def lv1(self):
        return self.N * self.K

    def bk_shape(self):
        return (self.n, (self.K + 1) * self.L, self.K + 1, self.N)

    def ks_shape(self):
        return (self.lv1, self.T, 1 << self.base_bit, self.n + 1)


def lib_path():
    return _LIB_FILE


def load_error():
    """None if the library loaded (or was not tried yet), else the exception text."""
    return _load_error


def library():
    """The loaded C library.  Raises TfheAmdError if libtfhe_amd.so is absent or unloadable:
    there is no fallback path."""
    global _lib, _load_error
    if _lib is not None:
        return _lib
    if not _LIB_FILE.exists():
        _load_error = f"{_LIB_FILE} not found: build it with `python -c 'import __graft_entry__ as g; g.build()'`"
        raise TfheAmdError(_load_error)
    try:
        # torch ships its own HIP runtime (same soname as the system one).  Two runtimes in one
        # process cannot both own the GPU, so when torch is installed it is imported first and
        # the library binds to the runtime torch has already loaded.
        try:
            import torch  # noqa: F401
        except ImportError:
            pass
        lib = C.CDLL(str(_LIB_FILE))
    except OSError as e:  # e.g. no HIP runtime on this machine
        _load_error = f"cannot load {_LIB_FILE}: {e}"
        raise TfheAmdError(_load_error) from e
    vp, sz, i32 = C.c_void_p, C.c_size_t, C.c_int
    pp = C.POINTER(Params)
    lib.tfhe_amd_params_default.argtypes = [pp]
    lib.tfhe_amd_params_default.restype = None
    lib.tfhe_amd_params_supported.argtypes = [pp]
    lib.tfhe_amd_bk_words.argtypes = [pp]
    lib.tfhe_amd_bk_words.restype = sz
    lib.tfhe_amd_ks_words.argtypes = [pp]
    lib.tfhe_amd_ks_words.restype = sz
    lib.tfhe_amd_ctx_create.argtypes = [pp, vp, vp, i32, C.c_uint, C.POINTER(vp)]
    lib.tfhe_amd_keygen.argtypes = [pp, vp, vp, C.c_double, C.c_double, vp, i32, vp, vp]
    lib.tfhe_amd_keygen_device.argtypes = [pp, vp, vp, C.c_double, C.c_double, vp, i32, vp, vp, vp]
    lib.tfhe_amd_ctx_create_from_secret.argtypes = [pp, vp, vp, C.c_double, C.c_double, vp, i32, C.c_uint,
                                                    C.POINTER(vp)]
    lib.tfhe_amd_ctx_destroy.argtypes = [vp]
    lib.tfhe_amd_ctx_destroy.restype = None
    lib.tfhe_amd_nand_batch.argtypes = [vp, vp, vp, vp, sz]
    lib.tfhe_amd_nand_batch_device.argtypes = [vp, vp, vp, vp, sz, vp]
    lib.tfhe_amd_nand_batch_device_noalloc.argtypes = [vp, vp, vp, vp, sz, vp]
    lib.tfhe_amd_gate_batch.argtypes = [vp, i32, vp, vp, vp, sz]
    lib.tfhe_amd_gate_batch_device.argtypes = [vp, i32, vp, vp, vp, sz, vp]
    lib.tfhe_amd_gates_batch_device.argtypes = [vp, vp, vp, vp, vp, sz, vp]
    lib.tfhe_amd_level_batch_device.argtypes = [vp, vp, vp, vp, vp, vp, sz, sz, vp]
    lib.tfhe_amd_not_batch.argtypes = [vp, vp, vp, sz]
    lib.tfhe_amd_not_batch_device.argtypes = [vp, vp, vp, sz, vp]
    lib.tfhe_amd_mux_batch.argtypes = [vp, vp, vp, vp, vp, sz]
    lib.tfhe_amd_mux_batch_device.argtypes = [vp, vp, vp, vp, vp, sz, vp]
    lib.tfhe_amd_bootstrap_batch.argtypes = [vp, vp, vp, sz]
    lib.tfhe_amd_bootstrap_batch_device.argtypes = [vp, vp, vp, sz, vp]
    lib.tfhe_amd_gate_bootstrapping_batch.argtypes = [vp, vp, vp, sz]
    lib.tfhe_amd_gate_bootstrapping_batch_device.argtypes = [vp, vp, vp, sz, vp]
    lib.tfhe_amd_blind_rotate_batch.argtypes = [vp, vp, vp, vp, sz]
    lib.tfhe_amd_blind_rotate_batch_device.argtypes = [vp, vp, vp, vp, sz, vp]
    lib.tfhe_amd_identity_key_switch_batch.argtypes = [vp, vp, vp, sz]
    lib.tfhe_amd_identity_key_switch_batch_device.argtypes = [vp, vp, vp, sz, vp]
    lib.tfhe_amd_external_product_batch.argtypes = [pp, vp, vp, vp, sz, i32]
    lib.tfhe_amd_cmux_batch.argtypes = [pp, vp, vp, vp, vp, sz, i32]
    lib.tfhe_amd_multi_ctx_create.argtypes = [pp, vp, vp, C.POINTER(i32), i32, C.c_uint, C.POINTER(vp)]
    lib.tfhe_amd_multi_ctx_destroy.argtypes = [vp]
    lib.tfhe_amd_multi_ctx_destroy.restype = None
    lib.tfhe_amd_multi_ctx_device_count.argtypes = [vp]
    lib.tfhe_amd_multi_ctx_get.argtypes = [vp, i32]
    lib.tfhe_amd_multi_ctx_get.restype = vp
    lib.tfhe_amd_multi_shard.argtypes = [vp, sz, i32, C.POINTER(sz), C.POINTER(sz)]
    lib.tfhe_amd_multi_shard.restype = None
    lib.tfhe_amd_multi_nand_batch.argtypes = [vp, vp, vp, vp, sz]
    lib.tfhe_amd_multi_gate_batch.argtypes = [vp, i32, vp, vp, vp, sz]
    lib.tfhe_amd_multi_bootstrap_batch.argtypes = [vp, vp, vp, sz]
    lib.tfhe_amd_multi_nand_batch_device.argtypes = [vp, C.POINTER(vp), C.POINTER(vp), C.POINTER(vp), C.POINTER(sz),
                                                     C.POINTER(vp)]
    lib.tfhe_amd_multi_ctx_synchronize.argtypes = [vp]
    lib.tfhe_amd_ctx_kernel_family.argtypes = [vp]
    lib.tfhe_amd_ctx_kernel_family.restype = C.c_char_p
    lib.tfhe_amd_ctx_synchronize.argtypes = [vp]
    lib.tfhe_amd_ctx_enable_timing.argtypes = [vp, i32]
    lib.tfhe_amd_ctx_timing.argtypes = [vp, C.POINTER(C.c_double), C.POINTER(C.c_double), C.POINTER(C.c_int64)]
    lib.tfhe_amd_last_error.restype = C.c_char_p
    lib.tfhe_amd_version.restype = C.c_char_p
    _lib = lib
    return lib


def version():
    return library().tfhe_amd_version().decode()


def params_default():
    p = Params()
    library().tfhe_amd_params_default(C.byref(p))
    return p


def _check(rc):
    if rc != 0:
        raise TfheAmdError(f"tfhe_amd error {rc}: {library().tfhe_amd_last_error().decode()}")


def _host_u32(a, shape=None):
    a = np.ascontiguousarray(a, dtype=np.uint32)
    if shape is not None and a.shape != tuple(shape):
        raise ValueError(f"expected shape {tuple(shape)}, got {a.shape}")
    return a


def _dev_ptr(t, words):
    """Address of a contiguous 4-byte-element torch tensor on the GPU holding `words` elements."""
    if not (t.is_cuda and t.is_contiguous() and t.element_size() == 4 and t.numel() == words):
        raise ValueError("expected a contiguous 32-bit GPU tensor of %d elements" % words)
    return C.c_void_p(t.data_ptr())


# Gate codes of the C ABI (TFHE_AMD_GATE_*).  ANDNY = (not a) and b, ANDYN = a and (not b).
GATES = {"nand": 0, "and": 1, "or": 2, "nor": 3, "xor": 4, "xnor": 5, "andny": 6, "andyn": 7, "orny": 8, "oryn": 9}


def gate_code(op):
    if isinstance(op, str):
        if op not in GATES:
            raise ValueError(f"unknown gate {op!r}")
        return GATES[op]
    return int(op)


class Context:
    """Device-resident evaluation key: what holding a HomNAND::EvaluateKey
    (reference src/hom_nand.hpp:36-42) becomes on the GPU."""

    KEYS_ON_DEVICE, GENERIC_KERNELS, LATENCY_FORM, THROUGHPUT_FORM, REFERENCE_FFT = 1, 2, 4, 8, 16

    def __init__(self, params, bk, ks, device=0, keys_on_device=False, generic_kernels=False, form=None,
                 reference_fft=False):
        """bk / ks: numpy uint32 arrays (host) or 32-bit torch GPU tensors (keys_on_device);
        either may be None.  generic_kernels: use the parameter-generic exact O(N^2) kernels
        even where a specialised NTT kernel exists.  form: None (by batch size), 'latency' or
'throughput' for the specialised blind rotation.  reference_fft: verification mode, products
        through a replay of the reference's FFT (output bit-identical to the compiled reference)."""
        lib = library()
        self.params = params
        self.device = int(device)
        self._h = C.c_void_p()
        flags = (self.KEYS_ON_DEVICE if keys_on_device else 0) | (self.GENERIC_KERNELS if generic_kernels else 0)
        flags |= {None: 0, 'latency': self.LATENCY_FORM, 'throughput': self.THROUGHPUT_FORM}[form]
        flags |= self.REFERENCE_FFT if reference_fft else 0
        bk_words, ks_words = lib.tfhe_amd_bk_words(C.byref(params)), lib.tfhe_amd_ks_words(C.byref(params))
        if keys_on_device:
            bk_p = _dev_ptr(bk, bk_words) if bk is not None else None
            ks_p = _dev_ptr(ks, ks_words) if ks is not None else None
        else:
            keep = []
            def host(x, words):
                if x is None:
                    return None
                x = _host_u32(x).reshape(-1)
                if x.size != words:
                    raise ValueError("key sizes do not match the parameter set")
                keep.append(x)
                return x.ctypes.data_as(C.c_void_p)
            bk_p, ks_p = host(bk, bk_words), host(ks, ks_words)
        _check(lib.tfhe_amd_ctx_create(C.byref(params), bk_p, ks_p, self.device, flags, C.byref(self._h)))

    @classmethod
    def from_secret(cls, params, s_lv0, s_ring, alpha_lv0, alpha_lv1, seed, device=0, form=None):
        """Evaluation keys generated on the GPU from a secret key (what make_evaluate_key does in
        the reference, src/hom_nand.hpp:56-64) and kept there.  s_lv0: [n] 0/1, s_ring: [K, N] 0/1."""
        s0, sr = _secret_bytes(params, s_lv0, s_ring)
        self = cls.__new__(cls)
        self.params, self.device, self._h = params, int(device), C.c_void_p()
        flags = {None: 0, 'latency': cls.LATENCY_FORM, 'throughput': cls.THROUGHPUT_FORM}[form]
        sw = seed_words(seed)
        _check(library().tfhe_amd_ctx_create_from_secret(
            C.byref(params), s0.ctypes.data_as(C.c_void_p), sr.ctypes.data_as(C.c_void_p), float(alpha_lv0),
            float(alpha_lv1), sw.ctypes.data_as(C.c_void_p), self.device, flags, C.byref(self._h)))
        return self

    def close(self):
        if getattr(self, "_h", None) is not None and self._h.value:
            library().tfhe_amd_ctx_destroy(self._h)
            self._h = C.c_void_p()

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def __enter__(self):
        return self

    def __exit__(self, *exc):
        self.close()

    # ---- host arrays -------------------------------------------------------
    def nand(self, a, b):
        """out[g] = NAND(a[g], b[g]); a, b: [G, n+1] numpy uint32 (reference src/hom_nand.hpp:74-81)."""
        a, b = _host_u32(a), _host_u32(b)
        if a.shape != b.shape or a.ndim != 2 or a.shape[1] != self.params.n1:
            raise ValueError("a and b must both be [G, n+1]")
        out = np.empty_like(a)
        _check(library().tfhe_amd_nand_batch(self._h, a.ctypes.data_as(C.c_void_p), b.ctypes.data_as(C.c_void_p),
                                             out.ctypes.data_as(C.c_void_p), a.shape[0]))
        return out

    def gate(self, op, a, b):
        """out[g] = op(a[g], b[g]) for a gate name or code: the bootstrap of NAND behind another
        linear combination (the reference has NAND only, src/hom_nand.hpp:76-78)."""
        a, b = _host_u32(a), _host_u32(b)
        if a.shape != b.shape or a.ndim != 2 or a.shape[1] != self.params.n1:
            raise ValueError("a and b must both be [G, n+1]")
        out = np.empty_like(a)
        _check(library().tfhe_amd_gate_batch(self._h, gate_code(op), a.ctypes.data_as(C.c_void_p),
                                             b.ctypes.data_as(C.c_void_p), out.ctypes.data_as(C.c_void_p), a.shape[0]))
        return out

    def not_(self, a):
        """out[g] = -a[g]: homomorphic NOT, no bootstrap."""
        a = _host_u32(a)
        if a.ndim != 2 or a.shape[1] != self.params.n1:
            raise ValueError("a must be [G, n+1]")
        out = np.empty_like(a)
        _check(library().tfhe_amd_not_batch(self._h, a.ctypes.data_as(C.c_void_p), out.ctypes.data_as(C.c_void_p),
                                            a.shape[0]))
        return out

    def mux(self, sel, x1, x0):
        """out[g] = sel[g] ? x1[g] : x0[g] (two blind rotations, one key switch per gate)."""
        sel, x1, x0 = _host_u32(sel), _host_u32(x1), _host_u32(x0)
        if not (sel.shape == x1.shape == x0.shape) or sel.ndim != 2 or sel.shape[1] != self.params.n1:
            raise ValueError("sel, x1 and x0 must all be [G, n+1]")
        out = np.empty_like(sel)
        _check(library().tfhe_amd_mux_batch(self._h, sel.ctypes.data_as(C.c_void_p), x1.ctypes.data_as(C.c_void_p),
                                            x0.ctypes.data_as(C.c_void_p), out.ctypes.data_as(C.c_void_p),
                                            sel.shape[0]))
        return out

    def bootstrap(self, c):
        """out[g] = identity_key_switch(gate_bootstrapping_tlwe_to_tlwe(c[g])) (reference src/hom_nand.hpp:79-80)."""
        c = _host_u32(c)
        if c.ndim != 2 or c.shape[1] != self.params.n1:
            raise ValueError("c must be [G, n+1]")
        out = np.empty_like(c)
        _check(library().tfhe_amd_bootstrap_batch(self._h, c.ctypes.data_as(C.c_void_p),
                                                  out.ctypes.data_as(C.c_void_p), c.shape[0]))
        return out

    def gate_bootstrapping_tlwe_to_tlwe(self, c):
        """lv1[g] = sample_extract_index(blind_rotate(c[g]), 0) (reference src/trgsw.hpp:172-182)."""
        c = _host_u32(c)
        lv1 = np.empty((c.shape[0], self.params.lv1 + 1), dtype=np.uint32)
        _check(library().tfhe_amd_gate_bootstrapping_batch(self._h, c.ctypes.data_as(C.c_void_p),
                                                           lv1.ctypes.data_as(C.c_void_p), c.shape[0]))
        return lv1

    def blind_rotate(self, c, test_vector=None):
        """acc[g] after the CMUX chain over c[g] (reference src/trgsw.hpp:131-170); test_vector
        [K+1, N] or None for the gate-bootstrapping one."""
        c = _host_u32(c)
        tv = None if test_vector is None else _host_u32(test_vector, (self.params.K + 1, self.params.N))
        acc = np.empty((c.shape[0], self.params.K + 1, self.params.N), dtype=np.uint32)
        _check(library().tfhe_amd_blind_rotate_batch(self._h, c.ctypes.data_as(C.c_void_p),
                                                     tv.ctypes.data_as(C.c_void_p) if tv is not None else None,
                                                     acc.ctypes.data_as(C.c_void_p), c.shape[0]))
        return acc

    def identity_key_switch(self, lv1):
        """out[g] = identity_key_switch(lv1[g]) (reference src/tlwe_key_switching.hpp:47-65)."""
        lv1 = _host_u32(lv1)
        out = np.empty((lv1.shape[0], self.params.n1), dtype=np.uint32)
        _check(library().tfhe_amd_identity_key_switch_batch(self._h, lv1.ctypes.data_as(C.c_void_p),
                                                            out.ctypes.data_as(C.c_void_p), lv1.shape[0]))
        return out

    # ---- GPU tensors (torch), asynchronous on `stream` (a torch stream or None) -------------
    OWN_STREAM = "own"

    def _stream_ptr(self, stream):
        """None -> torch's current stream on the context's GPU, so that the call is ordered behind
        whatever produced its tensors; Context.OWN_STREAM -> the context's own stream (the caller
        orders it against the producers); a torch stream -> that stream.  The default stream's
        handle is 0, which the C ABI spells TFHE_AMD_DEFAULT_STREAM = (void*)-1."""
        if stream == self.OWN_STREAM:
            return None
        if stream is None:
            import torch

            stream = torch.cuda.current_stream(self.device)
        handle = stream.cuda_stream
        return C.c_void_p(handle if handle else C.c_void_p(-1).value)

    def nand_device(self, a, b, out, stream=None):
        g = a.shape[0]
        w = g * self.params.n1
        _check(library().tfhe_amd_nand_batch_device(self._h, _dev_ptr(a, w), _dev_ptr(b, w), _dev_ptr(out, w), g,
                                                    self._stream_ptr(stream)))
        return out

    NOALLOC_SCRATCH_BYTES = 16 << 20

    def nand_device_noalloc(self, a, b, out, stream=None):
        """nand_device through the entry point that allocates nothing: the workspace is a static
        device array (NOALLOC_SCRATCH_BYTES per GPU, shared by every context on it), larger batches
        are rejected, and calls that overlap on one GPU race."""
        g = a.shape[0]
        w = g * self.params.n1
        _check(library().tfhe_amd_nand_batch_device_noalloc(self._h, _dev_ptr(a, w), _dev_ptr(b, w), _dev_ptr(out, w),
                                                            g, self._stream_ptr(stream)))
        return out

    def gate_device(self, op, a, b, out, stream=None):
        """op: a gate name or code for the whole batch, or a uint8 GPU tensor of codes, one per gate."""
        g = a.shape[0]
        w = g * self.params.n1
        if hasattr(op, "data_ptr"):
            if not (op.is_cuda and op.is_contiguous() and op.element_size() == 1 and op.numel() == g):
                raise ValueError("per-gate codes must be a contiguous uint8 GPU tensor of %d elements" % g)
            _check(library().tfhe_amd_gates_batch_device(self._h, C.c_void_p(op.data_ptr()), _dev_ptr(a, w),
                                                         _dev_ptr(b, w), _dev_ptr(out, w), g, self._stream_ptr(stream)))
        else:
            _check(library().tfhe_amd_gate_batch_device(self._h, gate_code(op), _dev_ptr(a, w), _dev_ptr(b, w),
                                                        _dev_ptr(out, w), g, self._stream_ptr(stream)))
        return out

    def level_device(self, ops, a, b, x0, out, n_mux, stream=None):
        """One dependency level: two-input gates with per-gate codes `ops` (uint8 GPU tensor,
        [G - n_mux]) first, then n_mux MUX gates (sel = a row, x1 = b row, x0 rows in `x0`,
        [n_mux, n+1]); all blind rotations in one launch."""
        g = a.shape[0]
        w = g * self.params.n1
        n_two = g - n_mux
        if n_two and not (ops is not None and ops.is_cuda and ops.is_contiguous() and ops.element_size() == 1
                          and ops.numel() == n_two):
            raise ValueError("ops must be a contiguous uint8 GPU tensor of %d elements" % n_two)
        _check(library().tfhe_amd_level_batch_device(
            self._h, C.c_void_p(ops.data_ptr()) if n_two else None, _dev_ptr(a, w), _dev_ptr(b, w),
            _dev_ptr(x0, n_mux * self.params.n1) if n_mux else None, _dev_ptr(out, w), g, n_mux,
            self._stream_ptr(stream)))
        return out

    def not_device(self, a, out, stream=None):
        g = a.shape[0]
        w = g * self.params.n1
        _check(library().tfhe_amd_not_batch_device(self._h, _dev_ptr(a, w), _dev_ptr(out, w), g,
                                                   self._stream_ptr(stream)))
        return out

    def mux_device(self, sel, x1, x0, out, stream=None):
        g = sel.shape[0]
        w = g * self.params.n1
        _check(library().tfhe_amd_mux_batch_device(self._h, _dev_ptr(sel, w), _dev_ptr(x1, w), _dev_ptr(x0, w),
                                                   _dev_ptr(out, w), g, self._stream_ptr(stream)))
        return out

    def bootstrap_device(self, c, out, stream=None):
        g = c.shape[0]
        w = g * self.params.n1
        _check(library().tfhe_amd_bootstrap_batch_device(self._h, _dev_ptr(c, w), _dev_ptr(out, w), g,
                                                         self._stream_ptr(stream)))
        return out

    def gate_bootstrapping_tlwe_to_tlwe_device(self, c, lv1, stream=None):
        g = c.shape[0]
        _check(library().tfhe_amd_gate_bootstrapping_batch_device(
            self._h, _dev_ptr(c, g * self.params.n1), _dev_ptr(lv1, g * (self.params.lv1 + 1)), g,
            self._stream_ptr(stream)))
        return lv1

    def identity_key_switch_device(self, lv1, out, stream=None):
        g = lv1.shape[0]
        _check(library().tfhe_amd_identity_key_switch_batch_device(
            self._h, _dev_ptr(lv1, g * (self.params.lv1 + 1)), _dev_ptr(out, g * self.params.n1), g,
            self._stream_ptr(stream)))
        return out

    def kernel_family(self):
        """'ntt512', 'ntt-workgroup', 'generic', 'reference-fft-wave', 'reference-fft-workgroup' or 'none'."""
        return library().tfhe_amd_ctx_kernel_family(self._h).decode()

    def synchronize(self):
        """Waits for the context's own stream and, when torch is loaded, for torch's current stream
        on the context's GPU (where the *_device calls go by default)."""
        _check(library().tfhe_amd_ctx_synchronize(self._h))
        import sys

        torch = sys.modules.get("torch")
        if torch is not None and torch.cuda.is_available():
            torch.cuda.current_stream(self.device).synchronize()

    def enable_timing(self, enable=True):
        _check(library().tfhe_amd_ctx_enable_timing(self._h, 1 if enable else 0))

    def timing(self):
        """(blind_rotate_ms, key_switch_ms, calls) summed since the previous query."""
        br, ks, n = C.c_double(), C.c_double(), C.c_int64()
        _check(library().tfhe_amd_ctx_timing(self._h, C.byref(br), C.byref(ks), C.byref(n)))
        return br.value, ks.value, n.value


class MultiContext:
    """Several GPUs of one node behind one handle (tfhe_amd_multi_ctx): the keys are uploaded once,
    copied device to device, and a batch is split in contiguous blocks over the device list.
    devices: GPU ids; an id may appear more than once (each entry gets its own context)."""

    def __init__(self, params, bk, ks, devices, keys_on_device=False, form=None):
        lib = library()
        self.params = params
        self.devices = [int(d) for d in devices]
        self._h = C.c_void_p()
        flags = (Context.KEYS_ON_DEVICE if keys_on_device else 0)
        flags |= {None: 0, 'latency': Context.LATENCY_FORM, 'throughput': Context.THROUGHPUT_FORM}[form]
        bk_words, ks_words = lib.tfhe_amd_bk_words(C.byref(params)), lib.tfhe_amd_ks_words(C.byref(params))
        if keys_on_device:
            bk_p, ks_p = _dev_ptr(bk, bk_words), _dev_ptr(ks, ks_words)
        else:
            bk, ks = _host_u32(bk).reshape(-1), _host_u32(ks).reshape(-1)
            if bk.size != bk_words or ks.size != ks_words:
                raise ValueError("key sizes do not match the parameter set")
            bk_p, ks_p = bk.ctypes.data_as(C.c_void_p), ks.ctypes.data_as(C.c_void_p)
        ids = (C.c_int * len(self.devices))(*self.devices)
        _check(lib.tfhe_amd_multi_ctx_create(C.byref(params), bk_p, ks_p, ids, len(self.devices), flags,
                                             C.byref(self._h)))

    def close(self):
        if getattr(self, "_h", None) is not None and self._h.value:
            library().tfhe_amd_multi_ctx_destroy(self._h)
            self._h = C.c_void_p()

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def __enter__(self):
        return self

    def __exit__(self, *exc):
        self.close()

    def device_count(self):
        return library().tfhe_amd_multi_ctx_device_count(self._h)

    def shard(self, n_gates, index):
        """[begin, end) of the gates entry `index` of the device list evaluates."""
        lo, hi = C.c_size_t(), C.c_size_t()
        library().tfhe_amd_multi_shard(self._h, n_gates, index, C.byref(lo), C.byref(hi))
        return lo.value, hi.value

    def kernel_family(self, index=0):
        return library().tfhe_amd_ctx_kernel_family(library().tfhe_amd_multi_ctx_get(self._h, index)).decode()

    def nand(self, a, b):
        """out[g] = NAND(a[g], b[g]) over all devices; host arrays [G, n+1]."""
        return self.gate("nand", a, b)

    def gate(self, op, a, b):
        a, b = _host_u32(a), _host_u32(b)
        if a.shape != b.shape or a.ndim != 2 or a.shape[1] != self.params.n1:
            raise ValueError("a and b must both be [G, n+1]")
        out = np.empty_like(a)
        _check(library().tfhe_amd_multi_gate_batch(self._h, gate_code(op), a.ctypes.data_as(C.c_void_p),
                                                   b.ctypes.data_as(C.c_void_p), out.ctypes.data_as(C.c_void_p),
                                                   a.shape[0]))
        return out

    def bootstrap(self, c):
        c = _host_u32(c)
        if c.ndim != 2 or c.shape[1] != self.params.n1:
            raise ValueError("c must be [G, n+1]")
        out = np.empty_like(c)
        _check(library().tfhe_amd_multi_bootstrap_batch(self._h, c.ctypes.data_as(C.c_void_p),
                                                        out.ctypes.data_as(C.c_void_p), c.shape[0]))
        return out

    OWN_STREAMS = "own"

    def nand_device(self, a, b, out, streams=None):
        """a, b, out: one GPU tensor [G_g, n+1] per entry of the device list, each on that entry's
        GPU.  Asynchronous.  streams: None = torch's current stream on every entry's GPU, so that
        each share is ordered behind whatever produced its tensors and ahead of later torch work on
        them; MultiContext.OWN_STREAMS = every context's own stream (the caller orders producers
        and consumers; synchronize() waits); or one torch stream per entry."""
        n = len(self.devices)
        if not (len(a) == len(b) == len(out) == n):
            raise ValueError("one tensor per device expected")
        VP, SZ = C.c_void_p * n, C.c_size_t * n
        g = [int(t.shape[0]) for t in a]
        w = [gi * self.params.n1 for gi in g]
        pa = VP(*[_dev_ptr(t, wi).value or 0 for t, wi in zip(a, w)])
        pb = VP(*[_dev_ptr(t, wi).value or 0 for t, wi in zip(b, w)])
        po = VP(*[_dev_ptr(t, wi).value or 0 for t, wi in zip(out, w)])
        if streams == self.OWN_STREAMS:
            ps = None
        else:
            import torch

            if streams is None:
                streams = [torch.cuda.current_stream(d) for d in self.devices]
            if len(streams) != n:
                raise ValueError("one stream per device expected")
            # the default stream's handle is 0, which the C ABI spells TFHE_AMD_DEFAULT_STREAM = (void*)-1
            ps = VP(*[(s.cuda_stream or C.c_void_p(-1).value) for s in streams])
        _check(library().tfhe_amd_multi_nand_batch_device(self._h, pa, pb, po, SZ(*g), ps))
        return out

    def synchronize(self):
        """Waits for every context's own stream and, when torch is loaded, for torch's current
        stream on every entry's GPU (where nand_device goes by default)."""
        _check(library().tfhe_amd_multi_ctx_synchronize(self._h))
        import sys

        torch = sys.modules.get("torch")
        if torch is not None and torch.cuda.is_available():
            for d in set(self.devices):
                torch.cuda.current_stream(d).synchronize()


def seed_words(seed):
    """The 256-bit key-generation seed as eight uint32 words: None = fresh OS entropy (what a real
    key needs); an int (little endian, for reproducible tests); 32 bytes; or eight words."""
    if seed is None:
        seed = os.urandom(32)
    if isinstance(seed, int):
        if seed < 0 or seed >> 256:
            raise ValueError("the seed is a 256-bit unsigned number")
        seed = seed.to_bytes(32, "little")
    if isinstance(seed, (bytes, bytearray)):
        if len(seed) != 32:
            raise ValueError("the seed is 32 bytes")
        return np.frombuffer(bytes(seed), dtype="<u4").astype(np.uint32)
    w = np.ascontiguousarray(seed, dtype=np.uint32).reshape(-1)
    if w.size != 8:
        raise ValueError("the seed is eight 32-bit words")
    return w


def _secret_bytes(params, s_lv0, s_ring):
    s0 = np.ascontiguousarray(s_lv0, dtype=np.uint8).reshape(-1)
    sr = np.ascontiguousarray(s_ring, dtype=np.uint8).reshape(-1)
    if s0.size != params.n or sr.size != params.K * params.N:
        raise ValueError("secret key sizes do not match the parameter set")
    if s0.max(initial=0) > 1 or sr.max(initial=0) > 1:
        raise ValueError("secret keys are binary")
    return s0, sr


def keygen(params, s_lv0, s_ring, alpha_lv0, alpha_lv1, seed, device=0, want_bk=True, want_ks=True):
    """(bk, ks) as host arrays, generated on the GPU (reference src/trgsw.hpp:119-129,
    src/tlwe_key_switching.hpp:26-45).  seed: see seed_words (None = OS entropy).  A key that is
    not wanted comes back as None."""
    s0, sr = _secret_bytes(params, s_lv0, s_ring)
    bk = np.empty(params.bk_shape(), dtype=np.uint32) if want_bk else None
    ks = np.empty(params.ks_shape(), dtype=np.uint32) if want_ks else None
    sw = seed_words(seed)
    _check(library().tfhe_amd_keygen(C.byref(params), s0.ctypes.data_as(C.c_void_p), sr.ctypes.data_as(C.c_void_p),
                                     float(alpha_lv0), float(alpha_lv1), sw.ctypes.data_as(C.c_void_p), int(device),
                                     bk.ctypes.data_as(C.c_void_p) if want_bk else None,
                                     ks.ctypes.data_as(C.c_void_p) if want_ks else None))
    return bk, ks


def external_product(params, trgsw, trlwe, device=0):
    """out[b] = trgsw[b] (x) trlwe[b] (reference src/trgsw.hpp:100-110); host arrays."""
    trgsw, trlwe = _host_u32(trgsw), _host_u32(trlwe)
    out = np.empty_like(trlwe)
    _check(library().tfhe_amd_external_product_batch(C.byref(params), trgsw.ctypes.data_as(C.c_void_p),
                                                     trlwe.ctypes.data_as(C.c_void_p), out.ctypes.data_as(C.c_void_p),
                                                     trlwe.shape[0], int(device)))
    return out


def cmux(params, trgsw, d0, d1, device=0):
    """out[b] = trgsw[b] (x) (d1[b] - d0[b]) + d0[b] (reference src/trgsw.hpp:112-117); host arrays."""
    trgsw, d0, d1 = _host_u32(trgsw), _host_u32(d0), _host_u32(d1)
    out = np.empty_like(d0)
    _check(library().tfhe_amd_cmux_batch(C.byref(params), trgsw.ctypes.data_as(C.c_void_p),
                                         d0.ctypes.data_as(C.c_void_p), d1.ctypes.data_as(C.c_void_p),
                                         out.ctypes.data_as(C.c_void_p), d0.shape[0], int(device)))
    return out
