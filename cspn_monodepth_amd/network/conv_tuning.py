"""MIOpen tuning database for the stock convolutions of config 5's host model.

The reference's trainer sets ``cudnn.benchmark = True`` (main.py:37): the vendor library then picks every convolution
algorithm by timing.  On ROCm that search (MIOpen's exhaustive find + solver tuning) takes ~8 minutes on a fresh machine
for this network and is lost with the machine.  ``miopen_db/`` holds what that search wrote on an MI355X — MIOpen's own
user find-db / perf-db text files (100 KB), produced by ``bench.py --workload train --conv-autotune on`` under
``MIOPEN_USER_DB_PATH`` (tools/make_miopen_db.sh) for the shapes of a 3-frame 228x304 fp32 shard.  With it the library's
immediate mode picks the tuned solvers at once: 28.3 instead of 36.6 ms per training step, and no find phase in the first
steps (~20 s).  Other shapes / library versions simply miss in the database and behave as without it.

Nothing here touches the convolutions themselves — they stay stock PyTorch-ROCm ops (north_star).
"""
import os
import shutil
import tempfile

__all__ = ["DB_DIR", "use_tuned_conv_db"]

DB_DIR = os.path.join(os.path.dirname(os.path.abspath(__file__)), "miopen_db")


def _private_dir(path):
    """`path` as a directory only this user can touch: created 0700 if missing, otherwise it must BE a real directory (not a
    symlink) owned by this user with no group / other access — a pre-planted directory or link in a shared location is refused."""
    try:
        os.mkdir(path, 0o700)
    except FileExistsError:
        pass
    st = os.lstat(path)
    import stat
    if not stat.S_ISDIR(st.st_mode) or st.st_uid != os.getuid() or (st.st_mode & 0o077):
        raise RuntimeError("%s exists but is not a private directory of this user (refusing to write a convolution database into it)" % path)
    return path


def _install(src, dst):
    """Copy `src` to `dst` only when `dst` does not exist, without following a link planted at `dst` and without a window in
    which a half-written file is visible: exclusive create of a temporary name, then rename."""
    if os.path.lexists(dst):
        return                                     # MIOpen appends what it learns: a database already in use is left alone
    tmp = "%s.tmp.%d" % (dst, os.getpid())
    fd = os.open(tmp, os.O_WRONLY | os.O_CREAT | os.O_EXCL | getattr(os, "O_NOFOLLOW", 0), 0o600)
    try:
        with os.fdopen(fd, "wb") as out, open(src, "rb") as inp:
            shutil.copyfileobj(inp, out)
        os.rename(tmp, dst)
    except BaseException:
        try:
            os.unlink(tmp)
        except OSError:
            pass
        raise


def use_tuned_conv_db(rank=0, force=False):
    """Point MIOpen at a private, writable copy of the shipped database (MIOpen appends what it learns; one copy per rank and
    job so that ranks do not write one file).  Must run before the process's first convolution.  A MIOPEN_USER_DB_PATH already
    in the environment wins unless ``force``.  Returns the directory in use, or None when nothing was changed.

    The copy lives under the user's cache directory ($XDG_CACHE_HOME or ~/.cache, created 0700 — NOT a predictable name in the
    world-writable temp dir, where another local user could pre-create the directory or plant symlinks: ADVICE r3), in a
    directory per rank — `miopen_db_rank<r>`: a plain `python train.py` finds on its next run what MIOpen learned on this one
    (a per-pid name made a new directory every run and never reused one) — and, when a scheduler names the job
    (SLURM_JOB_ID, or a TORCHELASTIC_RUN_ID other than torchrun's default literal "none"), per job as well, so that the ranks of a
    job restart onto the database they extended and two named jobs of one user never share live files.  Concurrent UNNAMED runs
    of one user on one machine share the per-rank files; MIOpen appends whole lines, and CSPN_MIOPEN_DB_TAG gives a run its own
    directory.  Files are installed only when missing (exclusive create + rename; existing files — what MIOpen appended on
    earlier runs — are kept).  Per-pid directories of earlier versions and the random-name fallbacks of this one are not created
    any more (the fallback, used only when no private cache directory can be had, lives under the temp dir and is removed at
    interpreter exit)."""
    if os.environ.get("MIOPEN_USER_DB_PATH") and not force:
        return None
    files = [f for f in os.listdir(DB_DIR) if f.endswith(".txt")] if os.path.isdir(DB_DIR) else []
    if not files:
        return None
    cache = os.environ.get("XDG_CACHE_HOME") or os.path.join(os.path.expanduser("~"), ".cache")
    try:
        os.makedirs(cache, mode=0o700, exist_ok=True)
        base = _private_dir(os.path.join(cache, "cspn_monodepth_amd"))
    except (OSError, RuntimeError):
        base = tempfile.mkdtemp(prefix="cspn_miopen_db_")            # no usable home: a fresh 0700 directory with a random name,
        import atexit                                                 # gone with the process (nothing to reuse it by)
        atexit.register(shutil.rmtree, base, True)
    job = os.environ.get("CSPN_MIOPEN_DB_TAG") or os.environ.get("SLURM_JOB_ID") or os.environ.get("TORCHELASTIC_RUN_ID") or ""
    if job == "none":                                                 # torchrun's default run id: not a name
        job = ""
    job = "".join(c if c.isalnum() or c in "-_." else "_" for c in job)[:64]
    dst = _private_dir(os.path.join(base, "miopen_db_%srank%d" % (job + "_" if job else "", int(rank))))
    for f in files:
        _install(os.path.join(DB_DIR, f), os.path.join(dst, f))
    os.environ["MIOPEN_USER_DB_PATH"] = dst
    return dst
