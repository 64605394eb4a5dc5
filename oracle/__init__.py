"""oracle -- TEST INFRASTRUCTURE ONLY (the parity checker).

ctypes front-ends for
  * ``libat_oracle.so``     -- our CPU restatement (oracle/at_oracle.c), and
  * ``_ref/libat_ref.so``   -- the REAL reference (src/alignment.h) compiled in
                               place behind oracle/ref_harness.c, when present.

Only tests/, bench.py's cpu_baseline leg and __graft_entry__.smoke() may import
this package.  The product (aligntools.c_amd) never does.
"""
import ctypes as C
import os
import subprocess

HERE = os.path.dirname(os.path.abspath(__file__))

GLOBAL, LOCAL, FIT, OVERLAP, EDIT = 0, 1, 2, 3, 4
MODE_NAMES = {"global": GLOBAL, "local": LOCAL, "fit": FIT, "overlap": OVERLAP, "edit": EDIT}
OP_MID, OP_LOW, OP_UPP, OP_JUMP = 0, 1, 2, 3


class Scoring(C.Structure):
    _fields_ = [("m", C.c_int), ("u", C.c_int), ("o", C.c_int), ("e", C.c_int), ("j", C.c_int),
                ("use_jump", C.c_int), ("sites", C.POINTER(C.c_int)), ("nsites", C.c_int)]


def build(force=False):
    """(Re)build the checkers with oracle/Makefile (gcc only)."""
    if force or not os.path.exists(os.path.join(HERE, "libat_oracle.so")) or \
            os.path.getmtime(os.path.join(HERE, "libat_oracle.so")) < os.path.getmtime(os.path.join(HERE, "at_oracle.c")):
        subprocess.run(["make", "-C", HERE, "libat_oracle.so"], check=True, capture_output=True)
    if os.path.exists("/root/reference/src/alignment.h"):
        subprocess.run(["make", "-C", HERE, "ref"], check=True, capture_output=True)


_port = None
_ref = None


def _load_port():
    global _port
    if _port is None:
        build()
        lib = C.CDLL(os.path.join(HERE, "libat_oracle.so"))
        lib.ato_align.restype = C.c_int
        lib.ato_align.argtypes = [C.c_int, C.c_char_p, C.c_int, C.c_char_p, C.c_int, C.POINTER(Scoring),
                                  C.POINTER(C.c_double), C.c_char_p, C.c_char_p, C.c_int, C.POINTER(C.c_int),
                                  C.POINTER(C.c_int), C.POINTER(C.c_int), C.POINTER(C.c_int),
                                  C.c_char_p, C.POINTER(C.c_int)]
        lib.ato_time_batch.restype = C.c_double
        lib.ato_time_batch.argtypes = [C.c_int, C.c_int, C.c_char_p, C.c_int, C.c_int, C.POINTER(Scoring),
                                       C.POINTER(C.c_double)]
        _port = lib
    return _port


def have_ref():
    return os.path.exists(os.path.join(HERE, "_ref", "libat_ref.so"))


def _load_ref():
    global _ref
    if _ref is None:
        lib = C.CDLL(os.path.join(HERE, "_ref", "libat_ref.so"))
        lib.ref_align.restype = C.c_int
        lib.ref_align.argtypes = [C.c_int, C.c_char_p, C.c_int, C.c_char_p, C.c_int,
                                  C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int,
                                  C.POINTER(C.c_int), C.c_int, C.POINTER(C.c_double),
                                  C.c_char_p, C.c_char_p, C.c_int, C.POINTER(C.c_int)]
        lib.ref_time_batch.restype = C.c_double
        lib.ref_time_batch.argtypes = [C.c_int, C.c_int, C.c_char_p, C.c_int, C.c_int,
                                       C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int,
                                       C.POINTER(C.c_int), C.c_int, C.POINTER(C.c_double)]
        _ref = lib
    return _ref


def _sites_arr(sites):
    sites = list(sites or [])
    arr = (C.c_int * max(1, len(sites)))(*sites)
    return arr, len(sites)


def align(mode, s1, s2, m=1, u=-2, o=-5, e=-1, j=-10, use_jump=False, sites=None):
    """Restatement.  Returns dict(rc, score, r1, r2, end_i, end_j, state, ops)."""
    lib = _load_port()
    if isinstance(s1, str):
        s1 = s1.encode()
    if isinstance(s2, str):
        s2 = s2.encode()
    arr, n = _sites_arr(sites)
    sc = Scoring(m, u, o, e, j, 1 if use_jump else 0, C.cast(arr, C.POINTER(C.c_int)), n)
    cap = len(s1) + len(s2) + 1
    r1 = C.create_string_buffer(cap)
    r2 = C.create_string_buffer(cap)
    ops = C.create_string_buffer(cap)
    score = C.c_double(0)
    rlen, ei, ej, st, nops = C.c_int(0), C.c_int(0), C.c_int(0), C.c_int(0), C.c_int(0)
    rc = lib.ato_align(mode, s1, len(s1), s2, len(s2), C.byref(sc), C.byref(score), r1, r2, cap,
                       C.byref(rlen), C.byref(ei), C.byref(ej), C.byref(st), ops, C.byref(nops))
    return dict(rc=rc, score=int(score.value) if rc == 0 else None,
                r1=r1.raw[:rlen.value].decode("latin1"), r2=r2.raw[:rlen.value].decode("latin1"),
                end_i=ei.value, end_j=ej.value, state=st.value, ops=bytes(ops.raw[:nops.value]))


def ref_align(mode, s1, s2, m=1, u=-2, o=-5, e=-1, j=-10, use_jump=False, sites=None):
    """The real reference (needs oracle/_ref).  Returns dict(rc, score, r1, r2)."""
    lib = _load_ref()
    if isinstance(s1, str):
        s1 = s1.encode()
    if isinstance(s2, str):
        s2 = s2.encode()
    arr, n = _sites_arr(sites)
    cap = len(s1) + len(s2) + 1
    r1 = C.create_string_buffer(cap)
    r2 = C.create_string_buffer(cap)
    score = C.c_double(0)
    rlen = C.c_int(0)
    rc = lib.ref_align(mode, s1, len(s1), s2, len(s2), m, u, o, e, j, 1 if use_jump else 0,
                       C.cast(arr, C.POINTER(C.c_int)), n, C.byref(score), r1, r2, cap, C.byref(rlen))
    return dict(rc=rc, score=int(score.value) if rc == 0 else None,
                r1=r1.raw[:rlen.value].decode("latin1"), r2=r2.raw[:rlen.value].decode("latin1"))


def time_batch(mode, blob, n, l1, l2, m, u, o, e, j=-10, use_jump=False, sites=None, kind="auto"):
    """Wall seconds for n fixed-shape pairs on ONE thread.  kind: reference|port|auto."""
    arr, ns = _sites_arr(sites)
    chk = C.c_double(0)
    if kind == "auto":
        kind = "reference" if have_ref() else "port"
    if kind == "reference":
        t = _load_ref().ref_time_batch(mode, n, blob, l1, l2, m, u, o, e, j, 1 if use_jump else 0,
                                       C.cast(arr, C.POINTER(C.c_int)), ns, C.byref(chk))
    else:
        sc = Scoring(m, u, o, e, j, 1 if use_jump else 0, C.cast(arr, C.POINTER(C.c_int)), ns)
        t = _load_port().ato_time_batch(mode, n, blob, l1, l2, C.byref(sc), C.byref(chk))
    return t, chk.value, kind
