"""Builds ``libmmk_hip.so`` (gfx950) in-tree with hipcc.

``python -m mimikit_amd.build`` or :func:`build`.  hipcc cross-compiles without
a GPU; the resulting shared object travels with the source tree.
"""
import hashlib
import os
import shutil
import subprocess
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
CSRC = os.path.join(HERE, "csrc")
LIB_PATH = os.path.join(HERE, "libmmk_hip.so")
SOURCES = ["kernels.hip", "linear.hip", "gemm.hip", "skinny.hip", "features.hip", "wavenet_plan.hip", "wavenet_persist.hip", "wavenet_chain.hip", "wavenet_lpipe.hip", "wavenet_spipe.hip", "wavenet_bpipe.hip", "wavenet_prefill.hip", "srnn_plan.hip",
           "srnn_bottom.hip", "srnn_gru.hip", "srnn_resident.hip", "lstm_step.hip", "lstm_seq.hip", "lstm_inproj.hip", "istft.hip", "spectral2048.hip", "s2s_plan.hip"]
# every header under csrc/ (a header missing from a hand-kept list once left a stale library behind) + the C ABI
HEADERS = sorted(f for f in os.listdir(CSRC) if f.endswith((".h", ".inc"))) + [os.path.join("..", "..", "include", "mmk.h")]
ARCH = "gfx950"


def _hipcc() -> str:
    for cand in (os.environ.get("HIPCC"), shutil.which("hipcc"), "/opt/rocm/bin/hipcc"):
        if cand and os.path.exists(cand):
            return cand
    raise RuntimeError("hipcc not found (set HIPCC or install ROCm under /opt/rocm)")


def source_digest() -> str:
    """sha256 over the names and contents of every translation unit and header the library is made of - compiled into the library
    (``mmk_build_digest``), so that a ``libmmk_hip.so`` found in the tree can be told from one built from other sources"""
    h = hashlib.sha256()
    for name in sorted(SOURCES + HEADERS):
        h.update(os.path.basename(name).encode() + b"\0")
        with open(os.path.join(CSRC, name), "rb") as f:
            h.update(f.read())
        h.update(b"\0")
    return h.hexdigest()[:32]


def library_digest(lib_path: str = LIB_PATH):
    """the digest a built library carries, or None (missing file, a library from before the digest); read out of the file, not through
    dlopen: a path that is loaded already would answer for the old image"""
    if not os.path.exists(lib_path):
        return None
    with open(lib_path, "rb") as f:
        blob = f.read()
    at = blob.find(b"mmk-source-digest:")
    if at < 0:
        return None
    return blob[at + 18:at + 50].split(b"\0")[0].decode(errors="replace")


def _stale() -> bool:
    # by CONTENT, not by modification time: a prebuilt library that travelled with the tree is used only if it was compiled from
    # exactly these sources
    return library_digest(LIB_PATH) != source_digest()


def build(force: bool = False, verbose: bool = False, diag: bool = False) -> str:
    """``diag``: the diagnostic build ``libmmk_hip_diag.so`` (-DMMK_DIAG: in-kernel phase stamps and the timing experiments of
    DESIGN.md; selected at import by MMK_DIAG_LIB=1) instead of the product library"""
    if diag:
        return _build(os.path.join(HERE, "libmmk_hip_diag.so"), os.path.join(HERE, "build_diag"), ["-DMMK_DIAG"], verbose)
    if not force and not _stale():
        return LIB_PATH
    return _build(LIB_PATH, os.path.join(HERE, "build"), [], verbose)


def _build(lib_path: str, obj_dir: str, defines, verbose: bool) -> str:
    """one build at a time per tree (a file lock: conftest, __graft_entry__.build and load_library may all ask for one), and the library is
    linked beside its final name and moved over it in one step - a process that has the old image mapped keeps its (unlinked) file"""
    import fcntl
    os.makedirs(obj_dir, exist_ok=True)
    with open(os.path.join(obj_dir, ".lock"), "w") as lock:
        fcntl.flock(lock, fcntl.LOCK_EX)
        try:
            return _build_locked(lib_path, obj_dir, defines, verbose)
        finally:
            fcntl.flock(lock, fcntl.LOCK_UN)


def _build_locked(lib_path: str, obj_dir: str, defines, verbose: bool) -> str:
    hipcc = _hipcc()
    objs = []
    procs = []
    for src in SOURCES:
        obj = os.path.join(obj_dir, src.replace(".hip", ".o"))
        objs.append(obj)
        extra = [f'-DMMK_SOURCE_DIGEST="{source_digest()}"'] if src == "kernels.hip" else []
        cmd = [hipcc, f"--offload-arch={ARCH}", "-O3", "-std=c++17", "-fPIC", *defines, *extra, "-c", os.path.join(CSRC, src), "-o", obj]
        if verbose:
            print(" ".join(cmd), file=sys.stderr)
        procs.append((src, subprocess.Popen(cmd, stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True)))
    for src, pr in procs:
        out, _ = pr.communicate()
        if pr.returncode != 0:
            raise RuntimeError(f"hipcc failed on {src}:\n{out}")
    tmp_path = f"{lib_path}.{os.getpid()}.tmp"
    cmd = [hipcc, f"--offload-arch={ARCH}", "-shared", "-fPIC", "-o", tmp_path] + objs
    res = subprocess.run(cmd, stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True)
    if res.returncode != 0:
        if os.path.exists(tmp_path):
            os.remove(tmp_path)
        raise RuntimeError(f"link failed:\n{res.stdout}")
    os.replace(tmp_path, lib_path)
    return lib_path


if __name__ == "__main__":
    print(build(force="--force" in sys.argv, verbose=True, diag="--diag" in sys.argv))
