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


def use_tuned_conv_db(rank=0, force=False):
    """Point MIOpen at a private, writable copy of the shipped database (MIOpen appends what it learns; one copy per rank so
    that ranks do not write one file).  Must run before the process's first convolution.  A MIOPEN_USER_DB_PATH already in
    the environment wins unless ``force``.  Returns the directory in use, or None when nothing was changed."""
    if os.environ.get("MIOPEN_USER_DB_PATH") and not force:
        return None
    files = [f for f in os.listdir(DB_DIR) if f.endswith(".txt")] if os.path.isdir(DB_DIR) else []
    if not files:
        return None
    dst = os.path.join(tempfile.gettempdir(), "cspn_miopen_db_%d_rank%d" % (os.getuid(), int(rank)))
    os.makedirs(dst, exist_ok=True)
    for f in files:
        shutil.copyfile(os.path.join(DB_DIR, f), os.path.join(dst, f))
    os.environ["MIOPEN_USER_DB_PATH"] = dst
    return dst
