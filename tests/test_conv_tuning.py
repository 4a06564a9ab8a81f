"""cspn_monodepth_amd.network.conv_tuning: the shipped MIOpen database is found, copied per rank, and never overrides a
database the user configured (host logic only, no GPU)."""
import os

from cspn_monodepth_amd.network import conv_tuning


def test_shipped_database_is_present_and_is_miopen_text():
    files = sorted(os.listdir(conv_tuning.DB_DIR))
    assert any(f.endswith(".ufdb.txt") for f in files) and any(f.endswith(".udb.txt") for f in files)
    for f in files:
        with open(os.path.join(conv_tuning.DB_DIR, f)) as fh:
            first = fh.readline()
        assert "=" in first and "NCHW" in first          # "<problem key>=<solver>:<time or parameters>;..."


def test_use_tuned_conv_db_copies_per_rank_and_respects_the_environment(monkeypatch):
    monkeypatch.delenv("MIOPEN_USER_DB_PATH", raising=False)
    d0 = conv_tuning.use_tuned_conv_db(rank=0)
    assert d0 and os.environ["MIOPEN_USER_DB_PATH"] == d0
    assert sorted(os.listdir(d0)) == sorted(f for f in os.listdir(conv_tuning.DB_DIR) if f.endswith(".txt"))
    assert conv_tuning.use_tuned_conv_db(rank=1) is None        # already configured (by the call above): left alone
    monkeypatch.setenv("MIOPEN_USER_DB_PATH", "/somewhere/else")
    assert conv_tuning.use_tuned_conv_db(rank=1) is None and os.environ["MIOPEN_USER_DB_PATH"] == "/somewhere/else"
    d1 = conv_tuning.use_tuned_conv_db(rank=1, force=True)
    assert d1 and d1 != d0 and os.environ["MIOPEN_USER_DB_PATH"] == d1


def test_database_directory_is_stable_across_runs_of_an_unnamed_job(monkeypatch, tmp_path):
    """without a scheduler's job id the directory must not be keyed by the pid (a new directory every run,
    nothing MIOpen learned ever reused); torchrun's default run id, the literal "none", is not a name either."""
    monkeypatch.setenv("XDG_CACHE_HOME", str(tmp_path / "cache"))
    for var in ("MIOPEN_USER_DB_PATH", "TORCHELASTIC_RUN_ID", "SLURM_JOB_ID", "CSPN_MIOPEN_DB_TAG"):
        monkeypatch.delenv(var, raising=False)
    d = conv_tuning.use_tuned_conv_db(rank=2)
    assert os.path.basename(d) == "miopen_db_rank2" and str(os.getpid()) not in d
    monkeypatch.delenv("MIOPEN_USER_DB_PATH")
    monkeypatch.setenv("TORCHELASTIC_RUN_ID", "none")
    assert conv_tuning.use_tuned_conv_db(rank=2) == d
    monkeypatch.delenv("MIOPEN_USER_DB_PATH")
    monkeypatch.setenv("CSPN_MIOPEN_DB_TAG", "exp 7")
    assert os.path.basename(conv_tuning.use_tuned_conv_db(rank=2)) == "miopen_db_exp_7_rank2"


def test_database_copy_is_private_and_never_follows_planted_links(monkeypatch, tmp_path):
    """the per-rank copy must not live under a predictable name in the world-writable temp dir, must refuse
    a directory it does not own privately, must not follow a symlink planted where a file goes, and must keep what MIOpen
    appended on an earlier run."""
    import stat
    import pytest
    monkeypatch.delenv("MIOPEN_USER_DB_PATH", raising=False)
    monkeypatch.setenv("XDG_CACHE_HOME", str(tmp_path / "cache"))
    monkeypatch.setenv("TORCHELASTIC_RUN_ID", "job/42")
    d = conv_tuning.use_tuned_conv_db(rank=3)
    assert d.startswith(str(tmp_path / "cache")) and "job_42" in d and d.endswith("rank3")
    assert stat.S_IMODE(os.lstat(d).st_mode) == 0o700
    f = sorted(os.listdir(d))[0]
    with open(os.path.join(d, f), "a") as fh:
        fh.write("learned=1\n")
    monkeypatch.delenv("MIOPEN_USER_DB_PATH")
    assert conv_tuning.use_tuned_conv_db(rank=3) == d                 # same job + rank: the same directory ...
    assert open(os.path.join(d, f)).read().endswith("learned=1\n")   # ... and the appended line is still there
    # a symlink planted where a database file goes is left alone, its target untouched
    victim = tmp_path / "victim"
    victim.write_text("precious")
    monkeypatch.delenv("MIOPEN_USER_DB_PATH")
    d2 = os.path.join(os.path.dirname(d), "miopen_db_job_42_rank4")
    os.mkdir(d2, 0o700)
    os.symlink(str(victim), os.path.join(d2, f))
    conv_tuning.use_tuned_conv_db(rank=4)
    assert victim.read_text() == "precious"
    # a directory other users can write to is refused
    monkeypatch.delenv("MIOPEN_USER_DB_PATH")
    d3 = os.path.join(os.path.dirname(d), "miopen_db_job_42_rank5")
    os.mkdir(d3, 0o777)
    os.chmod(d3, 0o777)
    with pytest.raises(RuntimeError, match="not a private directory"):
        conv_tuning.use_tuned_conv_db(rank=5)
