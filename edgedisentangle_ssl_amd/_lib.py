"""Loader / builder for libdisgat_hip.so (the C ABI declared in include/disgat_hip.h).

The library is built in-tree with hipcc for gfx950 and bound with ctypes: plain
pointers and sizes only, no torch types cross the boundary.  There is NO CPU or
PyTorch fallback: if the library is missing or a launcher reports an error the
caller gets a RuntimeError.
"""
import ctypes
import os
import shutil
import subprocess
from concurrent.futures import ThreadPoolExecutor

_HERE = os.path.dirname(os.path.abspath(__file__))
CSRC = os.path.join(_HERE, "csrc")
LIB_PATH = os.path.join(_HERE, "libdisgat_hip.so")
FLAGS_PATH = LIB_PATH + ".flags"        # the -D flags the installed library was compiled with (disgat_build_flags() reports the same)
SOURCES = ["api_common.hip", "edge_fwd.hip", "aux_score.hip", "edge_bwd.hip", "gemm_split.hip", "gemm_rs.hip", "gemm_planes.hip", "gemm_b2b.hip", "linear_skinny.hip",
           "optim.hip", "pair_sample.hip", "seg_tables.hip", "cls_loss.hip", "wgrad_small.hip"]
ARCH = "gfx950"
ABI_VERSION = 10        # csrc/api_common.hip: bumped whenever a launcher's argument list changes (round 3: dropout seed
                        # counter, padding labels / items, score-gradient strides, amax outputs, plane outputs; round 4: pair sampler, classification loss, small weight gradients)

_lib = None


def _hipcc():
    for cand in (os.environ.get("HIPCC"), shutil.which("hipcc"), "/opt/rocm/bin/hipcc"):
        if cand and os.path.exists(cand):
            return cand
    raise RuntimeError("hipcc not found: cannot build libdisgat_hip.so")


def _extra_flags():
    """DISGAT_HIPCC_FLAGS, normalised: what this process expects the library to have been compiled with."""
    return " ".join(os.environ.get("DISGAT_HIPCC_FLAGS", "").split())


def _stale():
    if not os.path.exists(LIB_PATH):
        return True
    try:
        if open(FLAGS_PATH).read() != _extra_flags():       # e.g. a diagnostic build a tools/ script left behind
            return True
    except OSError:
        return True
    t = os.path.getmtime(LIB_PATH)
    deps = [os.path.join(CSRC, f) for f in os.listdir(CSRC)] + [os.path.join(_HERE, "..", "include", "disgat_hip.h")]
    return any(os.path.getmtime(d) > t for d in deps if os.path.exists(d))


def build(force=False, verbose=False):
    """Compile every HIP source for gfx950 and link libdisgat_hip.so in-tree.  Serialised across
    processes with a file lock (N ranks of one node import the package at the same time)."""
    if not force and not _stale():
        return LIB_PATH
    import fcntl
    with open(os.path.join(_HERE, ".build.lock"), "w") as lock:
        fcntl.flock(lock, fcntl.LOCK_EX)
        try:
            if not force and not _stale():      # another process built it while we waited
                return LIB_PATH
            return _build_locked(verbose)
        finally:
            fcntl.flock(lock, fcntl.LOCK_UN)


def _build_locked(verbose):
    hipcc = _hipcc()
    objdir = os.path.join(_HERE, "build")
    os.makedirs(objdir, exist_ok=True)
    srcs = [s for s in SOURCES if os.path.exists(os.path.join(CSRC, s))]

    def one(src):
        obj = os.path.join(objdir, src.replace(".hip", ".o"))
        cmd = [hipcc, f"--offload-arch={ARCH}", "-O3", "-std=c++17", "-fPIC", "-fno-gpu-rdc",
               *_extra_flags().split(),          # e.g. -DNAME for same-box A/B builds (tools/)
               f'-DDISGAT_BUILD_FLAGS="{_extra_flags()}"',                 # reported by disgat_build_flags()
               "-c", os.path.join(CSRC, src), "-o", obj]
        r = subprocess.run(cmd, capture_output=True, text=True)
        if r.returncode != 0:
            raise RuntimeError(f"hipcc failed on {src}:\n{r.stderr[-4000:]}")
        if verbose and r.stderr.strip():
            print(r.stderr)
        return obj

    with ThreadPoolExecutor(max_workers=min(4, len(srcs))) as ex:
        objs = list(ex.map(one, srcs))
    tmp = LIB_PATH + ".tmp"
    r = subprocess.run([hipcc, f"--offload-arch={ARCH}", "-shared", "-fPIC", "-o", tmp, *objs],
                       capture_output=True, text=True)
    if r.returncode != 0:
        raise RuntimeError(f"link failed:\n{r.stderr[-4000:]}")
    os.replace(tmp, LIB_PATH)
    with open(FLAGS_PATH, "w") as f:        # what _stale() compares before the first dlopen (a loaded library cannot be replaced)
        f.write(_extra_flags())
    return LIB_PATH


_c = ctypes
_P = _c.c_void_p
_SIGS = {
    "disgat_abi_version": (_c.c_int, []),
    "disgat_last_error": (_c.c_char_p, []),
    "disgat_build_flags": (_c.c_char_p, []),
    "disgat_edge_fwd": (_c.c_int, [_c.c_int, _P, _c.c_int, _P, _c.c_int64, _c.c_int, _c.c_int, _c.c_int, _c.c_int,
                                   _P, _c.c_int, _P, _c.c_int, _P, _c.c_int, _P, _P, _P, _P, _P, _P, _c.c_int,
                                   _c.c_float, _c.c_uint64, _P, _P, _P, _P, _P, _P, _P]),
    "disgat_edge_combine": (_c.c_int, [_P, _P, _c.c_int, _c.c_int, _c.c_int, _P, _P, _P, _P, _c.c_int, _P, _P, _P, _P]),
    "disgat_aux_score": (_c.c_int, [_c.c_int, _P, _P, _c.c_int64, _c.c_int, _c.c_int, _c.c_int, _c.c_int, _c.c_int,
                                    _c.c_int, _P, _c.c_int, _P, _c.c_int, _P, _c.c_int, _P, _P, _P, _P]),
    "disgat_pair_loss": (_c.c_int, [_P, _c.c_int64, _c.c_int, _c.c_int, _P, _P, _P, _P, _P, _P, _P]),
    "disgat_pair_loss_bwd": (_c.c_int, [_P, _c.c_int64, _c.c_int, _c.c_int, _c.c_int, _P, _P, _P, _P, _P, _c.c_int, _P]),
    "disgat_weight_bound": (_c.c_int, [_P, _c.c_int64, _c.c_int64, _c.c_int64, _c.c_int, _c.c_int, _c.c_int, _P, _c.c_float, _c.c_float, _P, _P]),
    "disgat_wgrad_small": (_c.c_int, [_P, _c.c_int64, _c.c_int64, _P, _c.c_int64, _c.c_int64, _c.c_int, _c.c_int, _c.c_int, _c.c_int,
                                      _c.c_int, _P, _P, _P]),
    "disgat_cls_loss": (_c.c_int, [_P, _c.c_int64, _P, _c.c_int, _c.c_int64, _c.c_int, _c.c_double, _c.c_double, _P, _c.c_int64,
                                   _P, _P, _P, _P]),
    "disgat_cls_loss_bwd": (_c.c_int, [_P, _c.c_int64, _P, _c.c_int, _c.c_int64, _c.c_int, _P, _c.c_double, _P, _c.c_int64, _P]),
    "disgat_bwd_alpha": (_c.c_int, [_P, _c.c_int, _P, _c.c_int64, _c.c_int, _c.c_int, _P, _c.c_int, _P, _P, _P, _P, _P,
                                    _P, _c.c_int, _P, _c.c_int, _c.c_float, _c.c_uint64, _P, _P]),
    "disgat_seg_grad_att3": (_c.c_int, [_P, _c.c_int, _P, _P, _P, _c.c_int64, _c.c_int, _c.c_int, _c.c_int, _c.c_int,
                                        _P, _c.c_int, _P, _c.c_int, _P, _P, _c.c_int, _P, _c.c_int, _P, _P]),
    "disgat_seg_grad_sign": (_c.c_int, [_P, _c.c_int, _P, _P, _c.c_int64, _c.c_int64, _c.c_int, _c.c_int, _c.c_int, _c.c_int, _P,
                                        _P, _c.c_int, _P, _P, _c.c_int, _P, _c.c_int, _c.c_int, _P, _P, _P]),
    "disgat_seg_grad_hx": (_c.c_int, [_c.c_int, _P, _c.c_int, _P, _P, _P, _c.c_int64, _c.c_int, _c.c_int, _c.c_int,
                                      _c.c_int, _P, _c.c_int, _P, _c.c_int, _c.c_int, _P, _P]),
    "disgat_seg_sum": (_c.c_int, [_P, _c.c_int, _P, _P, _c.c_int64, _c.c_int64, _c.c_int, _c.c_int, _c.c_int, _P, _c.c_int, _c.c_int, _P, _P]),
    "disgat_seg_combine": (_c.c_int, [_P, _P, _c.c_int, _c.c_int, _P, _P, _c.c_int, _c.c_int, _P, _P]),
    "disgat_seg_tables": (_c.c_int, [_P, _P, _c.c_int64, _c.c_int, _c.c_int, _P, _P, _P, _P, _c.c_int, _P, _P, _c.c_int, _P, _P]),
    "disgat_gemm_split": (_c.c_int, [_P, _c.c_int64, _c.c_int64, _P, _P, _P, _c.c_int64, _c.c_int64, _P, _c.c_int64,
                                     _c.c_int64, _c.c_int, _c.c_int, _c.c_int, _c.c_int, _c.c_int, _c.c_float,
                                     _c.c_int, _P]),
    "disgat_gemm_f16x3": (_c.c_int, [_P, _c.c_int64, _c.c_int64, _P, _P, _P, _P, _P, _c.c_int64, _c.c_int64, _P,
                                     _c.c_int64, _c.c_int64, _c.c_int, _c.c_int, _c.c_int, _c.c_int, _c.c_int,
                                     _c.c_float, _P]),
    "disgat_gemm_f16x3_tn": (_c.c_int, [_P, _c.c_int64, _c.c_int64, _P, _c.c_int64, _c.c_int64, _P, _P, _P, _c.c_int, _c.c_int,
                                        _c.c_int, _c.c_int, _c.c_int, _P]),
    "disgat_split_f16": (_c.c_int, [_P, _c.c_int64, _c.c_int64, _c.c_int64, _c.c_int, _c.c_int, _c.c_int, _P, _P, _P]),
    "disgat_split_f16_rm": (_c.c_int, [_P, _c.c_int64, _c.c_int64, _c.c_int64, _c.c_int, _c.c_int, _c.c_int, _P, _P, _P]),
    "disgat_gemm_planes": (_c.c_int, [_P, _P, _c.c_int64, _c.c_int64, _P, _P, _P, _P, _P, _c.c_int64, _c.c_int64, _P,
                                      _c.c_int64, _c.c_int64, _P, _P, _c.c_int64, _c.c_int64, _P, _c.c_int, _c.c_int,
                                      _c.c_int, _c.c_int, _c.c_int, _c.c_float, _P]),
    "disgat_gemm_planes_logits": (_c.c_int, [_P, _P, _c.c_int64, _c.c_int64, _P, _P, _P, _P, _P, _c.c_int64, _c.c_int64, _P, _P, _P,
                                             _P, _P, _c.c_int, _c.c_int, _c.c_int, _c.c_int, _c.c_int, _c.c_int, _c.c_float, _P]),
    "disgat_proj_fuse": (_c.c_int, [_P, _P, _c.c_int64, _c.c_int64, _P, _P, _P, _P, _P, _P, _P, _P, _c.c_int64, _c.c_int, _c.c_int,
                                    _c.c_int, _c.c_int, _c.c_int, _c.c_int, _c.c_float, _P]),
    "disgat_split_planes": (_c.c_int, [_P, _c.c_int64, _c.c_int64, _c.c_int, _c.c_int, _c.c_int, _P, _P, _P, _c.c_int64,
                                       _c.c_int64, _P]),
    "disgat_planes_to_f32": (_c.c_int, [_P, _P, _c.c_int64, _c.c_int64, _c.c_int, _c.c_int, _c.c_int, _P, _P, _c.c_int64,
                                        _c.c_int64, _P]),
    "disgat_debug_stamps": (_c.c_int, [_P, _c.c_int]),
    "disgat_debug_stamps_rs": (_c.c_int, [_P, _c.c_int]),
    "disgat_debug_stamps_b2b": (_c.c_int, [_P, _c.c_int]),
    "disgat_amax": (_c.c_int, [_P, _c.c_int64, _c.c_int64, _c.c_int, _c.c_int, _c.c_int, _P, _P]),
    "disgat_act_bwd": (_c.c_int, [_P, _P, _P, _c.c_int64, _c.c_int, _c.c_float, _P, _P]),
    "disgat_linear_skinny": (_c.c_int, [_P, _c.c_int64, _c.c_int64, _c.c_int, _P, _c.c_int64, _P, _c.c_int, _P, _c.c_int64, _P]),
    "disgat_linear_skinny_wgrad": (_c.c_int, [_P, _c.c_int64, _c.c_int64, _c.c_int, _P, _c.c_int64, _c.c_int, _P, _c.c_int, _P]),
    "disgat_pair_sample_plan": (_c.c_int, [_P, _c.c_int, _c.c_int, _P, _c.c_int64, _c.c_int64, _c.c_double, _c.c_int64, _P, _P, _P, _P,
                                           _P, _P]),
    "disgat_pair_sample_emit": (_c.c_int, [_P, _c.c_int, _c.c_int, _P, _c.c_int64, _c.c_int64, _c.c_double, _c.c_int64, _c.c_int64,
                                           _c.c_int64, _P, _P, _P, _P, _P, _c.c_int, _P]),
    "disgat_adam_multi": (_c.c_int, [_c.c_int, _P, _P, _P, _P, _P, _P, _P, _P, _c.c_double, _c.c_double, _c.c_float, _P]),
    "disgat_adam_multi_dev": (_c.c_int, [_c.c_int, _P, _P, _P, _P, _P, _P, _P, _P, _c.c_double, _c.c_double, _c.c_float, _P]),
}


def exported_symbols():
    """Names include/disgat_hip.h declares (checked by the CPU test-suite)."""
    return sorted(_SIGS)


def load():
    """dlopen the in-tree library (building it first if sources are newer)."""
    global _lib
    if _lib is not None:
        return _lib
    if _stale():
        build()
    lib = ctypes.CDLL(LIB_PATH)
    lib.disgat_abi_version.restype = _c.c_int
    if lib.disgat_abi_version() != ABI_VERSION:
        # a library built from other sources than the signatures below describe (a stale in-tree .so whose timestamps look
        # fresh): calling it would pass arguments in the wrong slots - rebuild once, then refuse
        build(force=True)
        lib = ctypes.CDLL(LIB_PATH)
        lib.disgat_abi_version.restype = _c.c_int
        if lib.disgat_abi_version() != ABI_VERSION:
            raise RuntimeError(f"libdisgat_hip.so reports ABI {lib.disgat_abi_version()}, this package binds ABI {ABI_VERSION}")
    def flags_of(l):
        if not hasattr(l, "disgat_build_flags"):
            return None
        l.disgat_build_flags.restype = _c.c_char_p
        return l.disgat_build_flags().decode()

    if flags_of(lib) != _extra_flags():
        # e.g. a -DRS_DIAG / -DBB_DIAG build a tools/ script left behind: timestamps look fresh, but it is not the library
        # this process asked for - rebuild with this process's flags, then refuse
        build(force=True)
        lib = ctypes.CDLL(LIB_PATH)
        lib.disgat_abi_version.restype = _c.c_int
        if lib.disgat_abi_version() != ABI_VERSION or flags_of(lib) != _extra_flags():
            raise RuntimeError(f"libdisgat_hip.so was compiled with flags {flags_of(lib)!r}, "
                               f"this process expects {_extra_flags()!r} (DISGAT_HIPCC_FLAGS)")
    for name, (res, args) in _SIGS.items():
        try:
            fn = getattr(lib, name)
        except AttributeError:
            continue
        fn.restype = res
        fn.argtypes = args
    _lib = lib
    return lib


def call(name, *args):
    """Invoke a launcher; raise RuntimeError with the library's message on failure."""
    lib = load()
    rc = getattr(lib, name)(*args)
    if rc != 0:
        msg = lib.disgat_last_error()
        raise RuntimeError(f"{name} failed (rc={rc}): {msg.decode() if msg else ''}")
