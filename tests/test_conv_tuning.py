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
