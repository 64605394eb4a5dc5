"""aligntools.c_amd -- host-side mirror of the reference's function surface over
the gfx950 C-ABI shim (include/aligntools_hip.h, libaligntools_hip.so).

Mirrors reference src/alignment.h:
    opt_t / init_opt            :57-65, :102-114
    align_gla                   :417     align_local_affine   :805
    align_fit_affine_jump       :596     align_overlap        :926
    edit_dist                   :291
Every DP cell is computed by the HIP kernel; there is no CPU fallback.  If the
extension is not built, or no GPU is visible, the calls raise.
"""
import ctypes as C
import os

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.environ.get("AT_LIB_PATH") or os.path.join(HERE, "libaligntools_hip.so")   # (AT_LIB_PATH: A/B builds, tools/)

MODE_GLOBAL, MODE_LOCAL, MODE_FIT, MODE_OVERLAP, MODE_EDIT = 0, 1, 2, 3, 4
MODES = {"global": MODE_GLOBAL, "local": MODE_LOCAL, "fit": MODE_FIT, "overlap": MODE_OVERLAP, "edit": MODE_EDIT}
OP_MID, OP_LOW, OP_UPP, OP_JUMP = 0, 1, 2, 3
ST_LOW, ST_MID, ST_UPP = 1, 2, 3

# every symbol include/aligntools_hip.h declares
ABI_SYMBOLS = ["at_init", "at_destroy", "at_last_error", "at_set_scoring", "at_set_min_score", "at_align_batch",
               "at_align_batch_device", "at_align_allpairs_device", "at_render_batch_device", "at_compact_ops_device",
               "at_align_batch_strings", "at_align_allpairs", "at_align_allpairs_stream",
               "at_comm_init", "at_comm_broadcast_scoring", "at_comm_allgather", "at_comm_destroy", "at_comm_abi_checked",
               "at_pack_words", "at_pack_batch", "at_render", "at_last_config"]

_i32p = C.POINTER(C.c_int32)
_i64p = C.POINTER(C.c_int64)
_u8p = C.POINTER(C.c_uint8)
_u32p = C.POINTER(C.c_uint32)
_lib = None
# at_allpairs_chunk_fn of include/aligntools_hip.h
ALLPAIRS_CHUNK_FN = C.CFUNCTYPE(C.c_int, C.c_void_p, C.c_int64, C.c_int64, _i32p, _i32p, _i32p, _i32p)


class AlignToolsError(RuntimeError):
    def __init__(self, code, msg):
        super().__init__("aligntools error %d: %s" % (code, msg))
        self.code = code


def _one_hip_runtime():
    """One HIP runtime per process, whatever the import order.  PyTorch-ROCm bundles its own libamdhip64.so (same soname
    as the system one the shim is linked against); a process that loads the system runtime first and torch's second ends up
    with two, and torch then sees no GPU.  If torch is installed, its runtime is loaded here (by path, without importing
    torch) before the shim: the shim's DT_NEEDED entry then resolves to it, and a later `import torch` finds it already
    there.  Without torch (the C host, the CLI) the system runtime is used."""
    import importlib.util
    try:
        spec = importlib.util.find_spec("torch")
    except (ImportError, ValueError):
        spec = None
    if spec is None or not spec.submodule_search_locations:
        return
    for name in ("libamdhip64.so", "libamdhip64.so.7", "libamdhip64.so.6"):
        cand = os.path.join(list(spec.submodule_search_locations)[0], "lib", name)
        if os.path.exists(cand):
            try:
                C.CDLL(cand, mode=C.RTLD_GLOBAL)
            except OSError:
                pass
            return


def load_library():
    """Load the shim.  Raises if it has not been built (no silent fallback)."""
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.exists(LIB_PATH):
        raise ImportError("aligntools.c_amd: %s is missing -- run `python -c 'import __graft_entry__ as g; g.build()'` "
                          "or `make -C aligntools/c_amd`; there is no CPU fallback" % LIB_PATH)
    _one_hip_runtime()
    lib = C.CDLL(LIB_PATH)
    lib.at_init.restype = C.c_int
    lib.at_init.argtypes = [C.POINTER(C.c_int), C.c_int, C.POINTER(C.c_void_p)]
    lib.at_destroy.restype = None
    lib.at_destroy.argtypes = [C.c_void_p]
    lib.at_last_error.restype = C.c_char_p
    lib.at_last_error.argtypes = [C.c_void_p]
    lib.at_last_config.restype = C.c_char_p
    lib.at_last_config.argtypes = [C.c_void_p]
    lib.at_set_scoring.restype = C.c_int
    lib.at_set_scoring.argtypes = [C.c_void_p] + [C.c_int] * 6 + [C.POINTER(C.c_int), C.c_int]
    lib.at_set_min_score.restype = C.c_int
    lib.at_set_min_score.argtypes = [C.c_void_p, C.c_int, C.c_int32]
    lib.at_align_batch.restype = C.c_int
    lib.at_align_batch.argtypes = [C.c_void_p, C.c_int, C.c_int64, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p,
                                   C.c_void_p, C.c_int, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p,
                                   C.c_void_p, C.c_void_p, C.c_void_p]
    lib.at_align_batch_device.restype = C.c_int
    lib.at_align_batch_device.argtypes = [C.c_void_p, C.c_int, C.c_int64, C.c_void_p, C.c_int, C.c_void_p, C.c_void_p,
                                          C.c_void_p, C.c_void_p, C.c_int32, C.c_int32, C.c_int, C.c_int,
                                          C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p,
                                          C.c_void_p, C.c_void_p]
    lib.at_align_allpairs_device.restype = C.c_int
    lib.at_align_allpairs_device.argtypes = [C.c_void_p, C.c_int, C.c_int64, C.c_void_p, C.c_int, C.c_void_p, C.c_void_p,
                                             C.c_int32, C.c_int64, C.c_int64, C.c_int, C.c_void_p, C.c_void_p,
                                             C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p]
    lib.at_render_batch_device.restype = C.c_int
    lib.at_render_batch_device.argtypes = [C.c_void_p, C.c_int64, C.c_void_p, C.c_int] + [C.c_void_p] * 10 + [C.c_int, C.c_void_p]
    lib.at_compact_ops_device.restype = C.c_int
    lib.at_compact_ops_device.argtypes = [C.c_void_p, C.c_int64, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_int64,
                                          C.c_void_p, C.c_void_p]
    lib.at_align_batch_strings.restype = C.c_int
    lib.at_align_batch_strings.argtypes = [C.c_void_p, C.c_int, C.c_int64] + [C.c_void_p] * 13
    lib.at_align_allpairs.restype = C.c_int
    lib.at_align_allpairs.argtypes = [C.c_void_p, C.c_int, C.c_int64, C.c_void_p, C.c_void_p, C.c_void_p, C.c_int64, C.c_int64, C.c_int,
                                      C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p]
    lib.at_align_allpairs_stream.restype = C.c_int
    lib.at_align_allpairs_stream.argtypes = [C.c_void_p, C.c_int, C.c_int64, C.c_void_p, C.c_void_p, C.c_void_p, C.c_int64, C.c_int64, C.c_int64,
                                             ALLPAIRS_CHUNK_FN, C.c_void_p]
    lib.at_comm_abi_checked.restype = C.c_int
    lib.at_comm_abi_checked.argtypes = []
    lib.at_pack_words.restype = C.c_int64
    lib.at_pack_words.argtypes = [C.c_int64, C.c_void_p, C.c_void_p, C.c_int]
    lib.at_pack_batch.restype = C.c_int
    lib.at_pack_batch.argtypes = [C.c_int64, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_int,
                                  C.POINTER(C.c_int), C.c_void_p, C.c_void_p, C.c_void_p]
    lib.at_render.restype = C.c_int
    lib.at_render.argtypes = [C.c_void_p, C.c_int32, C.c_void_p, C.c_int32, C.c_void_p, C.c_int32, C.c_void_p, C.c_void_p]
    _lib = lib
    return lib


def kernel_source_sha16():
    """Fingerprint of the SWEEP kernels' sources -- at_sweep.hip.h, at_sweep16.hip.h, at_walk16.hip.h, at_myers.hip.h, at_launch.h and the instantiation
    units at_k*.hip / at_myers.hip; not the packing / rendering kernels nor the host-side plumbing (at_hip.hip, at_comm.hip), whose
    choices show in the kernel configuration string: the committed PMC counters under profiles/ belong to one state of those
    kernels, and bench.py prices its roofline with them only while this fingerprint still matches."""
    import hashlib
    d = os.path.join(HERE, "csrc")
    h = hashlib.sha256()
    for name in sorted(os.listdir(d)):
        if name in ("at_sweep.hip.h", "at_sweep16.hip.h", "at_walk16.hip.h", "at_myers.hip.h", "at_launch.h", "at_myers.hip") or (name.startswith("at_k") and name.endswith(".hip")):
            h.update(name.encode() + b"\0")
            h.update(open(os.path.join(d, name), "rb").read())
    return h.hexdigest()[:16]


class opt_t:
    """The reference's scoring block with its defaults (init_opt, alignment.h:102-114)."""

    def __init__(self, m=1, u=-2, o=-5, e=-1, j=-10, s=False, sites=None):
        self.m, self.u, self.o, self.e, self.j, self.s = m, u, o, e, j, bool(s)
        self.sites = list(sites or [])


def init_opt():
    return opt_t()


def _ptr(a):
    return a.ctypes.data_as(C.c_void_p) if a is not None else None


def pack_pairs(pairs, bits=0):
    """Host helper: list of (s1, s2) bytes -> (words, woff1, woff2, len1, len2, bits)."""
    lib = load_library()
    blob, off1, len1, off2, len2 = _flatten(pairs)
    n = len(pairs)
    b = C.c_int(0)
    rc = lib.at_pack_batch(n, _ptr(blob), _ptr(off1), _ptr(len1), _ptr(off2), _ptr(len2), bits, C.byref(b), None,
                           _ptr(off1), _ptr(off2))
    if rc:
        raise AlignToolsError(rc, lib.at_last_error(None).decode())
    nw = lib.at_pack_words(n, _ptr(len1), _ptr(len2), b.value)
    words = np.zeros(nw, dtype=np.uint32)
    woff1 = np.zeros(n, dtype=np.int64)
    woff2 = np.zeros(n, dtype=np.int64)
    rc = lib.at_pack_batch(n, _ptr(blob), _ptr(off1), _ptr(len1), _ptr(off2), _ptr(len2), b.value, None,
                           _ptr(words), _ptr(woff1), _ptr(woff2))
    if rc:
        raise AlignToolsError(rc, lib.at_last_error(None).decode())
    return words, woff1, woff2, len1, len2, b.value


def _flatten(pairs):
    n = len(pairs)
    len1 = np.fromiter((len(p[0]) for p in pairs), dtype=np.int32, count=n)
    len2 = np.fromiter((len(p[1]) for p in pairs), dtype=np.int32, count=n)
    tot = len1.astype(np.int64) + len2
    starts = np.zeros(n, dtype=np.int64)
    if n > 1:
        np.cumsum(tot[:-1], out=starts[1:])
    off1 = starts
    off2 = starts + len1
    blob = np.frombuffer(b"".join(p[0] + p[1] for p in pairs) + b"\0", dtype=np.uint8).copy()
    return blob, off1, len1, off2, len2


def _b(s):
    return s.encode("latin1") if isinstance(s, str) else bytes(s)


class Aligner:
    """One handle = one GPU (one process per GPU)."""

    def __init__(self, device=None):
        self._lib = load_library()
        self._h = C.c_void_p()
        ids = None if device is None else (C.c_int * 1)(int(device))
        rc = self._lib.at_init(ids, 1 if device is not None else 0, C.byref(self._h))
        if rc:
            raise AlignToolsError(rc, self._lib.at_last_error(None).decode())

    def close(self):
        if self._h:
            self._lib.at_destroy(self._h)
            self._h = C.c_void_p()

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def _check(self, rc):
        if rc:
            raise AlignToolsError(rc, self._lib.at_last_error(self._h).decode())

    @property
    def last_config(self):
        return self._lib.at_last_config(self._h).decode()

    def set_scoring(self, m=1, u=-2, o=-5, e=-1, j=-10, use_jump=False, sites=None):
        sites = list(sites or [])
        arr = (C.c_int * max(1, len(sites)))(*sites)
        self._check(self._lib.at_set_scoring(self._h, m, u, o, e, j, 1 if use_jump else 0, arr, len(sites)))

    def set_opt(self, opt):
        self.set_scoring(opt.m, opt.u, opt.o, opt.e, opt.j, opt.s, opt.sites)

    def set_min_score(self, min_score=None):
        """All-vs-all overlap scores: pairs proven to score below `min_score` are not swept (state 0, score = an upper bound);
        None switches the threshold off (at_set_min_score)."""
        self._check(self._lib.at_set_min_score(self._h, 0 if min_score is None else 1, 0 if min_score is None else int(min_score)))

    def align_batch_strings(self, mode, pairs):
        """pairs: list of (s1, s2) bytes/str.  The two gapped strings of every pair, rendered on the GPU
        (at_align_batch_strings).  Returns a dict: score, end_i, end_j, state, nops (numpy) and r1, r2 (lists of str)."""
        if isinstance(mode, str):
            mode = MODES[mode]
        pairs = [(_b(a), _b(b)) for a, b in pairs]
        n = len(pairs)
        blob, off1, len1, off2, len2 = _flatten(pairs)
        score, end_i, end_j, state, nops = (np.zeros(n, dtype=np.int32) for _ in range(5))
        str_off = off1 + np.arange(n, dtype=np.int64)          # slots of l1+l2+1 bytes
        total = int(len(blob) + n + 64)
        r1 = np.zeros(total, dtype=np.uint8)
        r2 = np.zeros(total, dtype=np.uint8)
        self._check(self._lib.at_align_batch_strings(self._h, mode, n, _ptr(blob), _ptr(off1), _ptr(len1), _ptr(off2),
                                                     _ptr(len2), _ptr(score), _ptr(end_i), _ptr(end_j), _ptr(state),
                                                     _ptr(r1), _ptr(r2), _ptr(str_off), _ptr(nops)))
        b1, b2 = r1.tobytes(), r2.tobytes()
        out = dict(score=score, end_i=end_i, end_j=end_j, state=state, nops=nops)
        out["r1"] = [b1[str_off[k]:str_off[k] + nops[k]].decode("latin1") for k in range(n)]
        out["r2"] = [b2[str_off[k]:str_off[k] + nops[k]].decode("latin1") for k in range(n)]
        return out

    def render_batch_device(self, npairs, d_seq, bits, d_woff1, d_woff2, d_end_i, d_end_j, d_ops, d_ops_off, d_nops,
                            d_r1, d_r2, d_str_off=None, nul_terminate=False, stream=0):
        """Raw device-pointer entry of the rendering kernel (at_render_batch_device)."""
        self._check(self._lib.at_render_batch_device(self._h, npairs, d_seq, bits, d_woff1, d_woff2, d_end_i, d_end_j,
                                                     d_ops, d_ops_off, d_nops, d_r1, d_r2, d_str_off,
                                                     1 if nul_terminate else 0, stream))

    def compact_ops_device(self, npairs, d_ops, d_ops_off, d_nops, d_packed, packed_cap, d_packed_off, stream=0):
        """Slots of ops -> one contiguous payload + exclusive offsets (at_compact_ops_device)."""
        self._check(self._lib.at_compact_ops_device(self._h, npairs, d_ops, d_ops_off, d_nops, d_packed, packed_cap,
                                                    d_packed_off, stream))

    def align_batch(self, mode, pairs, traceback=True, render=True):
        """pairs: list of (s1, s2) bytes/str.  Returns a dict of numpy arrays
        (score, end_i, end_j, state, nops), `ops` (list of bytes, END->START) and,
        with render, `r1`/`r2` (the reference's two gapped strings, rendered on the host by at_render;
        align_batch_strings renders them on the GPU)."""
        if isinstance(mode, str):
            mode = MODES[mode]
        pairs = [(_b(a), _b(b)) for a, b in pairs]
        n = len(pairs)
        blob, off1, len1, off2, len2 = _flatten(pairs)
        tb = bool(traceback) and mode != MODE_EDIT
        score = np.zeros(n, dtype=np.int32)
        end_i = np.zeros(n, dtype=np.int32)
        end_j = np.zeros(n, dtype=np.int32)
        state = np.zeros(n, dtype=np.int32)
        nops = np.zeros(n, dtype=np.int32)
        ops_off = off1.copy()          # l1+l2 bytes per pair, same offsets as the blob
        ops = np.zeros(len(blob) + 64, dtype=np.uint8) if tb else None
        self._check(self._lib.at_align_batch(self._h, mode, n, _ptr(blob), _ptr(off1), _ptr(len1), _ptr(off2), _ptr(len2),
                                             1 if tb else 0, _ptr(score), _ptr(end_i), _ptr(end_j), _ptr(state),
                                             _ptr(ops), _ptr(ops_off) if tb else None, _ptr(nops) if tb else None))
        out = dict(score=score, end_i=end_i, end_j=end_j, state=state, nops=nops)
        if tb:
            out["ops"] = [bytes(ops[ops_off[k]:ops_off[k] + nops[k]]) for k in range(n)]
            if render:
                r1s, r2s = [], []
                for k in range(n):
                    a, b = self.render(out["ops"][k], pairs[k][0], int(end_i[k]), pairs[k][1], int(end_j[k]))
                    r1s.append(a)
                    r2s.append(b)
                out["r1"], out["r2"] = r1s, r2s
        return out

    def render(self, ops, s1, end_i, s2, end_j):
        n = len(ops)
        r1 = C.create_string_buffer(n + 1)
        r2 = C.create_string_buffer(n + 1)
        rc = self._lib.at_render(ops, n, s1, end_i, s2, end_j, r1, r2)
        if rc:
            raise AlignToolsError(rc, "at_render: ops inconsistent with the sequences")
        return r1.raw[:n].decode("latin1"), r2.raw[:n].decode("latin1")

    # raw device-pointer entry (bench.py): all arguments are integer device addresses
    def align_batch_device(self, mode, npairs, d_seq, bits, d_woff1, d_len1, d_woff2, d_len2, max_len1, max_len2,
                           uniform_shape, want_traceback, d_score, d_end_i, d_end_j, d_state, d_ops, d_ops_off, d_nops, stream=0):
        self._check(self._lib.at_align_batch_device(self._h, mode, npairs, d_seq, bits, d_woff1, d_len1, d_woff2, d_len2,
                                                    max_len1, max_len2, 1 if uniform_shape else 0,
                                                    1 if want_traceback else 0, d_score, d_end_i,
                                                    d_end_j, d_state, d_ops, d_ops_off, d_nops, stream))


    def align_allpairs_stream(self, mode, reads_blob, off, lens, first_pair, npairs, chunk_pairs, on_slice):
        """All-vs-all scores in slices with bounded memory (at_align_allpairs_stream): reads_blob uint8, off int64, lens int32
        (numpy); on_slice(first, score, end_i, end_j, state) gets numpy views that are valid during the call only."""
        if isinstance(mode, str):
            mode = MODES[mode]
        err = []

        def cb(_user, first, n, sc, ei, ej, st):
            try:
                on_slice(int(first), *(np.ctypeslib.as_array(x, shape=(int(n),)) for x in (sc, ei, ej, st)))
                return 0
            except BaseException as ex:   # (an exception must not unwind through the C frames)
                err.append(ex)
                return 1
        fn = ALLPAIRS_CHUNK_FN(cb)
        rc = self._lib.at_align_allpairs_stream(self._h, mode, len(lens), _ptr(reads_blob), _ptr(off), _ptr(lens), first_pair, npairs,
                                                chunk_pairs, fn, None)
        if err:
            raise err[0]
        self._check(rc)

    def align_allpairs_device(self, mode, nreads, d_seq, bits, d_woff, d_len, max_len, first_pair, npairs, want_traceback,
                              d_score, d_end_i, d_end_j, d_state, d_ops, d_ops_off, d_nops, stream=0):
        """All ordered pairs (a < b) of one read set; see at_align_allpairs_device in include/aligntools_hip.h."""
        self._check(self._lib.at_align_allpairs_device(self._h, mode, nreads, d_seq, bits, d_woff, d_len, max_len, first_pair,
                                                       npairs, 1 if want_traceback else 0, d_score, d_end_i, d_end_j, d_state,
                                                       d_ops, d_ops_off, d_nops, stream))


_default = None


def default_aligner():
    global _default
    if _default is None:
        _default = Aligner()
    return _default


def _single(mode, s1, s2, opt):
    al = default_aligner()
    al.set_opt(opt)
    r = al.align_batch(mode, [(s1, s2)])
    if mode == MODE_EDIT:
        return int(r["score"][0])
    return float(r["score"][0]), r["r1"][0], r["r2"][0]


# ---- the reference's five kernels, same names and argument meaning (opt_t) ----
def align_gla(s1, s2, opt=None):
    """Global affine alignment (reference alignment.h:417).  Returns (score, r1, r2)."""
    return _single(MODE_GLOBAL, s1, s2, opt or opt_t())


def align_local_affine(s1, s2, opt=None):
    """Smith-Waterman affine (reference alignment.h:805).  Returns (score, r1, r2)."""
    return _single(MODE_LOCAL, s1, s2, opt or opt_t())


def align_fit_affine_jump(s1, s2, opt=None):
    """Fit alignment, jump state when opt.s (reference alignment.h:596).  Raises
    like the reference dies when s1 is longer than s2 (:599)."""
    return _single(MODE_FIT, s1, s2, opt or opt_t())


def align_overlap(s1, s2, opt=None):
    """Overlap alignment, linear gap (reference alignment.h:926)."""
    return _single(MODE_OVERLAP, s1, s2, opt or opt_t())


def edit_dist(s1, s2, opt=None):
    """Edit distance with `-u` as the signed mismatch cost (reference alignment.h:291)."""
    return _single(MODE_EDIT, s1, s2, opt or opt_t())
