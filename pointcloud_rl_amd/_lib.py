"""ctypes binding of libpcrl_hip.so (the C ABI declared in include/pcrl.h).

The library is built in-tree by ``__graft_entry__.build()`` / ``make -C pointcloud_rl_amd/csrc``.
There is no CPU fallback: if the shared object is missing, loading raises.
"""
import ctypes
import os

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.environ.get("PCRL_HIP_LIB") or os.path.join(_HERE, "libpcrl_hip.so")     # env override: A/B builds during kernel work

PCRL_MAX_SEG = 4
PCRL_MAX_CHANNELS = 16
DT_F32, DT_U8, DT_BOOL = 0, 1, 2
AUG_JITTER, AUG_AFFINE, AUG_SUBSAMPLE, AUG_COLOR = 1, 2, 4, 8


class FeatSeg(ctypes.Structure):
    _fields_ = [("ptr", ctypes.c_void_p), ("dtype", ctypes.c_int32), ("channels", ctypes.c_int32),
                ("div255", ctypes.c_int32), ("_pad", ctypes.c_int32),
                ("stride_b", ctypes.c_int64), ("stride_c", ctypes.c_int64), ("stride_n", ctypes.c_int64)]


class CloudDesc(ctypes.Structure):
    _fields_ = [("B", ctypes.c_int32), ("N", ctypes.c_int32), ("nseg", ctypes.c_int32), ("row_div", ctypes.c_int32),
                ("seg", FeatSeg * PCRL_MAX_SEG)]


class AugDesc(ctypes.Structure):
    _fields_ = [("flags", ctypes.c_int32), ("row_mul", ctypes.c_int32), ("row_add", ctypes.c_int32), ("_pad", ctypes.c_int32),
                ("jitter_noise", ctypes.c_void_p),
                ("jitter_lo", ctypes.c_float), ("jitter_hi", ctypes.c_float),
                ("seed", ctypes.c_uint64), ("offset", ctypes.c_uint64), ("affine", ctypes.c_void_p),
                ("offset_ptr", ctypes.c_void_p), ("point_index", ctypes.c_void_p), ("n_index", ctypes.c_int32), ("color_order", ctypes.c_int32),
                ("color_factor", ctypes.c_float * 4), ("color_one_minus", ctypes.c_float * 4), ("color_mean", ctypes.c_void_p),
                ("n_index_ptr", ctypes.c_void_p)]


class EncoderWeights(ctypes.Structure):
    _fields_ = [("c_in", ctypes.c_int32), ("c1", ctypes.c_int32), ("c2", ctypes.c_int32), ("c3", ctypes.c_int32),
                ("w0", ctypes.c_void_p), ("b0", ctypes.c_void_p), ("w1", ctypes.c_void_p), ("g1", ctypes.c_void_p),
                ("be1", ctypes.c_void_p), ("w2", ctypes.c_void_p), ("g2", ctypes.c_void_p), ("be2", ctypes.c_void_p),
                ("eps", ctypes.c_float), ("_pad", ctypes.c_int32)]


class AdamPending(ctypes.Structure):
    _fields_ = [("partial", ctypes.c_void_p), ("n_partial", ctypes.c_int32), ("_pad", ctypes.c_int32),
                ("grad_norm_out", ctypes.c_void_p), ("step_counter", ctypes.c_void_p)]


class AdamRider(ctypes.Structure):
    _fields_ = [("param", ctypes.c_void_p), ("grad", ctypes.c_void_p), ("exp_avg", ctypes.c_void_p), ("exp_avg_sq", ctypes.c_void_p),
                ("n", ctypes.c_size_t), ("lr", ctypes.c_float), ("beta1", ctypes.c_float), ("beta2", ctypes.c_float), ("eps", ctypes.c_float),
                ("grad_scale", ctypes.c_float), ("_pad", ctypes.c_int32), ("step_counter", ctypes.c_void_p), ("grad_norm_out", ctypes.c_void_p),
                ("partial", ctypes.c_void_p)]


class ColGather(ctypes.Structure):
    _fields_ = [("src", ctypes.c_void_p), ("head_stride", ctypes.c_int64), ("heads", ctypes.c_int32), ("rows", ctypes.c_int32),
                ("ld", ctypes.c_int32), ("col0", ctypes.c_int32), ("ncols", ctypes.c_int32), ("_pad", ctypes.c_int32), ("dst", ctypes.c_void_p)]


class LnJob(ctypes.Structure):
    _fields_ = [("x", ctypes.c_void_p), ("ldx", ctypes.c_int64), ("M", ctypes.c_int32), ("n_dst", ctypes.c_int32),
                ("dst", ctypes.c_void_p * 4), ("ld_dst", ctypes.c_int64 * 4), ("xhat", ctypes.c_void_p), ("rstd", ctypes.c_void_p),
                ("cat_src", ctypes.c_void_p * 2), ("cat_dst", ctypes.c_void_p * 2), ("cat_ld_src", ctypes.c_int64 * 2),
                ("cat_ld_dst", ctypes.c_int64 * 2), ("cat_n", ctypes.c_int32 * 2), ("cat_row_div", ctypes.c_int32 * 2)]


class FeatureHead(ctypes.Structure):
    _fields_ = [("weight", ctypes.c_void_p), ("bias", ctypes.c_void_p), ("gamma", ctypes.c_void_p), ("beta", ctypes.c_void_p),
                ("F", ctypes.c_int32), ("eps", ctypes.c_float), ("n_ranges", ctypes.c_int32), ("begin", ctypes.c_int32 * 2),
                ("job", LnJob * 2)]


class ColsumJob(ctypes.Structure):
    _fields_ = [("part", ctypes.c_void_p), ("blk_stride", ctypes.c_int64), ("nblk", ctypes.c_int32), ("ncols", ctypes.c_int32),
                ("out", ctypes.c_void_p), ("scale", ctypes.c_float), ("op", ctypes.c_int32)]


class GatherSeg(ctypes.Structure):
    _fields_ = [("src", ctypes.c_void_p), ("dst", ctypes.c_void_p), ("row_bytes", ctypes.c_int64)]


class GemmDesc(ctypes.Structure):
    _fields_ = [("A", ctypes.c_void_p), ("B", ctypes.c_void_p), ("C", ctypes.c_void_p), ("bias", ctypes.c_void_p), ("mask", ctypes.c_void_p),
                ("M", ctypes.c_int32), ("N", ctypes.c_int32), ("K", ctypes.c_int32), ("batch", ctypes.c_int32),
                ("a_stride_m", ctypes.c_int64), ("a_stride_k", ctypes.c_int64), ("b_stride_k", ctypes.c_int64), ("b_stride_n", ctypes.c_int64),
                ("ldc", ctypes.c_int64), ("ld_mask", ctypes.c_int64),
                ("a_batch_stride", ctypes.c_int64), ("b_batch_stride", ctypes.c_int64), ("c_batch_stride", ctypes.c_int64),
                ("bias_batch_stride", ctypes.c_int64), ("mask_batch_stride", ctypes.c_int64),
                ("relu", ctypes.c_int32), ("ones_col", ctypes.c_int32), ("accumulate", ctypes.c_int32), ("_pad", ctypes.c_int32),
                ("C_ones", ctypes.c_void_p), ("c_ones_batch_stride", ctypes.c_int64)]


class PcrlError(RuntimeError):
    pass


_lib = None
RUNTIME_LINKS = os.path.join(_HERE, "_hiprt")


def _needed_hip_soname():
    """The name under which libpcrl_hip.so asks the loader for the HIP runtime (its DT_NEEDED entry, e.g. libamdhip64.so.7), read from the
    built library itself; None when it cannot be read (library not built, no binutils)."""
    import re
    import subprocess
    if not os.path.exists(LIB_PATH):
        return None
    for tool in ("readelf", "/opt/rocm/lib/llvm/bin/llvm-readelf"):
        try:
            out = subprocess.run([tool, "-d", LIB_PATH], capture_output=True, text=True, timeout=30).stdout
        except (OSError, subprocess.SubprocessError):
            continue
        m = re.search(r"\(NEEDED\)[^\[]*\[(libamdhip64\.so[^\]]*)\]", out)
        if m:
            return m.group(1)
    return None


def ensure_runtime_links():
    """Create pointcloud_rl_amd/_hiprt (machine-local, idempotent): symlinks to the HIP runtime torch's wheel bundles, which
    libpcrl_hip.so searches first (RPATH $ORIGIN/_hiprt, csrc/Makefile) -- so that a process holds ONE HIP runtime whether it maps
    the library before or after `import torch`.  Without torch installed nothing is created and the library uses /opt/rocm's.
    torch is located, not imported.  The link that satisfies the library's DT_NEEDED carries the name the LIBRARY asks for (read from its
    dynamic section) and is only made when torch's bundled runtime reports the same soname -- a wheel bundling another ROCm major must
    not be handed to a `.so.N` request.  Returns the directory or None."""
    import importlib.util
    spec = importlib.util.find_spec("torch")
    if spec is None or not spec.origin:
        return None
    src = os.path.join(os.path.dirname(spec.origin), "lib")
    bundled = os.path.join(src, "libamdhip64.so")
    if not os.path.exists(bundled):
        return None
    # fast path of every later process start: the stamp written with the links names the library build and the torch runtime they were
    # made for -- no readelf child processes when nothing changed
    stamp_path = os.path.join(RUNTIME_LINKS, ".stamp")
    try:
        stamp = f"{os.stat(LIB_PATH).st_mtime_ns}\n{os.path.realpath(bundled)}\n"
    except OSError:
        stamp = None
    try:
        with open(stamp_path) as f:
            have = f.read()
        if stamp and have.startswith(stamp):
            cached = have[len(stamp):].strip()
            if cached and os.path.realpath(os.path.join(RUNTIME_LINKS, cached)) == os.path.realpath(bundled):
                return RUNTIME_LINKS
    except OSError:
        pass
    needed = _needed_hip_soname() or "libamdhip64.so.7"
    if needed != "libamdhip64.so":
        # torch's copy must BE that ABI: its own soname (when readable) has to match the request
        import re
        import subprocess
        try:
            out = subprocess.run(["readelf", "-d", bundled], capture_output=True, text=True, timeout=60).stdout
            m = re.search(r"\(SONAME\)[^\[]*\[([^\]]*)\]", out)
            if m and m.group(1) != needed:
                return None
        except (OSError, subprocess.SubprocessError):
            pass
    marker = os.path.join(RUNTIME_LINKS, needed)
    fresh = lambda: os.path.islink(marker) and os.path.realpath(marker) == os.path.realpath(bundled)
    if fresh():
        return RUNTIME_LINKS
    # `_hiprt` is a symbolic link to a directory of links built aside; putting it in place (or replacing a stale one: another torch
    # installation) is ONE rename of the link (os.replace is atomic), so the ranks of a multi-GPU launch that all get here at once on a
    # fresh machine, and every loader resolving $ORIGIN/_hiprt meanwhile, see the old set or the new one -- never a missing or half-made
    # directory.  A real directory left by earlier versions of this function cannot be replaced by a link in one step: it is renamed aside
    # first (the only non-atomic case, once per machine that has one).
    import shutil
    target = f"{RUNTIME_LINKS}.d.{os.getpid()}"
    shutil.rmtree(target, ignore_errors=True)
    os.makedirs(target)
    for name in sorted(os.listdir(src)):
        if ".so" in name and os.path.isfile(os.path.join(src, name)):
            os.symlink(os.path.join(src, name), os.path.join(target, name))
    if not os.path.lexists(os.path.join(target, needed)):
        os.symlink(bundled, os.path.join(target, needed))
    if stamp:
        with open(os.path.join(target, ".stamp"), "w") as f:
            f.write(stamp + needed + "\n")
    link_tmp = f"{RUNTIME_LINKS}.lnk.{os.getpid()}"
    try:
        if os.path.lexists(link_tmp):
            os.unlink(link_tmp)
        os.symlink(os.path.basename(target), link_tmp)
        stale = os.path.realpath(RUNTIME_LINKS) if os.path.islink(RUNTIME_LINKS) else None
        if os.path.isdir(RUNTIME_LINKS) and not os.path.islink(RUNTIME_LINKS):
            aside = f"{RUNTIME_LINKS}.old.{os.getpid()}"
            os.rename(RUNTIME_LINKS, aside)
            shutil.rmtree(aside, ignore_errors=True)
        os.replace(link_tmp, RUNTIME_LINKS)
        if stale and stale != os.path.realpath(target) and os.path.basename(stale).startswith(os.path.basename(RUNTIME_LINKS) + ".d."):
            shutil.rmtree(stale, ignore_errors=True)
    except OSError:
        if os.path.lexists(link_tmp):
            os.unlink(link_tmp)
        if not fresh():
            shutil.rmtree(target, ignore_errors=True)
    return RUNTIME_LINKS if fresh() else None


class _TracedLib:
    """Debug switch PCRL_TRACE_LAUNCHES=<directory>: every pcrl_* entry point called through this binding writes `> name` to
    <directory>/rank<RANK>_pid<pid>.trace BEFORE the call and `< name rc` after the device has finished it (a stream synchronisation
    after every call unless PCRL_TRACE_SYNC=0 or the stream is capturing).  A rank that dies of an asynchronous GPU fault -- the queue
    abort callback calls abort() from a runtime thread -- leaves the name of the launch in flight as the file's last line; a last line
    `< name` says the fault came from work that is NOT this library's (torch, the collective's staging copies).  Written with os.write
    on an O_APPEND descriptor: nothing is buffered in the process."""

    def __init__(self, cdll, directory):
        os.makedirs(directory, exist_ok=True)
        self.__dict__["_cdll"] = cdll
        self.__dict__["_fd"] = os.open(os.path.join(directory, f"rank{os.environ.get('RANK', '0')}_pid{os.getpid()}.trace"),
                                       os.O_WRONLY | os.O_CREAT | os.O_APPEND, 0o644)
        self.__dict__["_sync"] = os.environ.get("PCRL_TRACE_SYNC", "1") != "0"
        self.__dict__["_wrapped"] = {}

    def __getattr__(self, name):
        fn = getattr(self._cdll, name)
        if not name.startswith("pcrl_") or name == "pcrl_last_error":
            return fn
        w = self._wrapped.get(name)
        if w is None:
            import torch
            fd, sync = self._fd, self._sync
            enter, leave = f"> {name}\n".encode(), f"< {name} ".encode()

            def w(*args):
                os.write(fd, enter)
                rc = fn(*args)
                if sync and torch.cuda.is_available() and not torch.cuda.is_current_stream_capturing():
                    torch.cuda.synchronize()
                os.write(fd, leave + str(rc).encode() + b"\n")
                return rc
            self._wrapped[name] = w
        return w

    def __setattr__(self, name, value):
        setattr(self._cdll, name, value)


def lib():
    """Load libpcrl_hip.so once.  Raises if it has not been built -- the product has no fallback path."""
    global _lib
    if _lib is None:
        if not os.path.exists(LIB_PATH):
            raise PcrlError(
                f"{LIB_PATH} is missing: build the HIP extension first "
                "(python -c 'import __graft_entry__ as g; g.build()' or make -C pointcloud_rl_amd/csrc)")
        # torch first: its wheel carries its own libamdhip64, and the process must end up with ONE HIP runtime.  With this
        # library (linked against /opt/rocm's) loaded before torch, kernels launched here fail with "no ROCm-capable device
        # is detected" while torch's own work runs (seen with build() followed by smoke() in one process on the GPU box).
        import torch  # noqa: F401
        try:
            ensure_runtime_links()       # for embedders that map the library themselves, possibly before torch
        except OSError:
            pass                         # a read-only tree: the torch-first order above is what keeps this process on one runtime
        _lib = ctypes.CDLL(LIB_PATH)
        _lib.pcrl_last_error.restype = ctypes.c_char_p
        if os.environ.get("PCRL_TRACE_LAUNCHES"):
            _lib = _TracedLib(_lib, os.environ["PCRL_TRACE_LAUNCHES"])
    return _lib


def check(rc):
    if rc != 0:
        raise PcrlError(f"pcrl error {rc}: {lib().pcrl_last_error().decode(errors='replace')}")


if __name__ == "__main__":
    import sys
    if "--link-runtime" in sys.argv:
        print("HIP runtime links:", ensure_runtime_links() or "torch not found: libpcrl_hip.so will use /opt/rocm's runtime")
