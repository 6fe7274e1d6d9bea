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
import atexit
import hashlib
import os
import shutil
import tarfile
import tempfile

_HERE = os.path.dirname(os.path.abspath(__file__))
BUNDLE = os.path.join(_HERE, "_ref", "reference_py.tgz")
_cached = None


def _ok(root):
    return bool(root) and os.path.isdir(os.path.join(root, "FantasyWorld", "fusion"))


def _trusted(root, tag):
    """An extraction of bundle `tag` that this user made: owned by us, not writable by group / others, complete."""
    if not _ok(root):
        return False
    st = os.stat(root)
    if st.st_uid != os.getuid() or (st.st_mode & 0o022):
        return False
    try:
        with open(os.path.join(root, ".fw_bundle")) as f:
            return f.read().strip() == tag
    except OSError:
        return False


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
        # Extracted into a PER-USER directory of mode 0700 (never a predictable path in the shared temp dir that another user of the
        # machine could pre-plant: the test process imports what it finds there), member paths checked by tarfile's "data" filter,
        # and a directory found there is only trusted when this user owns it, nobody else can write it, and it carries the marker
        # written after a COMPLETE extraction of THIS bundle (its hash).
        base = os.path.join(os.environ.get("XDG_CACHE_HOME") or os.path.join(os.path.expanduser("~"), ".cache"), "fw_reference")
        dst = os.path.join(base, tag)
        if not _trusted(dst, tag):
            os.makedirs(base, mode=0o700, exist_ok=True)
            os.chmod(base, 0o700)
            tmp = tempfile.mkdtemp(prefix=f"{tag}_x_", dir=base)
            with tarfile.open(BUNDLE, "r:gz") as tar:
                tar.extractall(tmp, filter="data")
            with open(os.path.join(tmp, ".fw_bundle"), "w") as f:
                f.write(tag)
            try:
                os.rename(tmp, dst)
            except OSError:                 # another process of this user won the race, or a stale / untrusted directory is in the way
                if _trusted(dst, tag):
                    shutil.rmtree(tmp, ignore_errors=True)          # the winner's copy is good: drop ours
                else:
                    # an untrusted leftover (partial extraction, wrong mode): move it aside when it is ours and retry once, so the cache
                    # heals instead of growing by one bundle copy per run
                    try:
                        if os.lstat(dst).st_uid == os.getuid():
                            aside = tempfile.mkdtemp(prefix=f"{tag}_stale_", dir=base)
                            os.rename(dst, os.path.join(aside, "d"))
                            shutil.rmtree(aside, ignore_errors=True)
                            os.rename(tmp, dst)
                    except OSError:
                        pass
                    if not _trusted(dst, tag):
                        dst = tmp           # use the private extraction itself, and remove it when this process ends
                        atexit.register(shutil.rmtree, tmp, ignore_errors=True)
        if _trusted(dst, tag):
            _cached = dst
            return dst
    _cached = ""
    return None


def available():
    return reference_root() is not None
