"""TEST INFRASTRUCTURE ONLY -- where the REAL reference lives on the machine a test / the CPU-baseline leg runs on.

Resolution order:
  1. $FW_REFERENCE_ROOT                      (explicit)
  2. /root/reference                          (the build container)
  3. oracle/_ref/reference_py.tgz             (the GPU box: the UNMODIFIED reference Python packages, staged by
                                               oracle/stage_ref.sh in the build container; oracle/_ref/ is git-ignored, so the
                                               bundle is never committed, but it travels with the gpurun snapshot like the built
                                               .so files do.  Extracted once per machine into a scratch directory.)
Nothing under fantasy_world_amd/ imports this file; with none of the three present the callers skip (tests) or fall back to the
oracle restatement (bench.py's cpu_baseline leg, kind = "port")."""
import hashlib
import os
import tarfile
import tempfile

_HERE = os.path.dirname(os.path.abspath(__file__))
BUNDLE = os.path.join(_HERE, "_ref", "reference_py.tgz")
_cached = None


def _ok(root):
    return bool(root) and os.path.isdir(os.path.join(root, "FantasyWorld", "fusion"))


def reference_root():
    """-> directory holding `FantasyWorld/` (importable as a package root), or None."""
    global _cached
    if _cached is not None:
        return _cached or None
    for cand in (os.environ.get("FW_REFERENCE_ROOT"), "/root/reference"):
        if _ok(cand):
            _cached = cand
            return cand
    if os.path.isfile(BUNDLE):
        with open(BUNDLE, "rb") as f:
            tag = hashlib.sha1(f.read()).hexdigest()[:12]
        dst = os.path.join(tempfile.gettempdir(), f"fw_reference_{tag}")
        if not _ok(dst):
            tmp = tempfile.mkdtemp(prefix="fw_reference_x_")
            with tarfile.open(BUNDLE, "r:gz") as tar:
                tar.extractall(tmp)
            try:
                os.rename(tmp, dst)
            except OSError:                 # another process won the race
                pass
        if _ok(dst):
            _cached = dst
            return dst
    _cached = ""
    return None


def available():
    return reference_root() is not None
