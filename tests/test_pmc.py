"""CPU: the counter-pass plumbing of bench.py (chameleonrt_amd/pmc.py) -- parsing a rocprofv3 rocpd
database into per-kernel sums, and failing soft (an error entry, never an exception) when rocprofv3 or
the GPU is not there, so that `roofline.traffic` becomes null instead of the bench line disappearing."""
import sqlite3

from chameleonrt_amd import pmc


def test_read_db_sums_per_kernel(tmp_path):
    db = sqlite3.connect(tmp_path / "x_results.db")
    db.execute("create table kernels (name text, start integer, end integer)")
    db.execute("create table counters_collection (kernel_name text, counter_name text, value real)")
    rows = [("void crt::k_trace_closest<true, false>(crt::SceneView, int)", 0, 5_000_000),
            ("void crt::k_trace_closest<true, false>(crt::SceneView, int)", 10_000_000, 13_000_000),
            ("crt::k_shade(crt::SceneView)", 0, 1_000_000)]
    db.executemany("insert into kernels values (?, ?, ?)", rows)
    db.executemany("insert into counters_collection values (?, ?, ?)",
                   [(rows[0][0], "FETCH_SIZE", 100.0), (rows[0][0], "FETCH_SIZE", 50.0), (rows[2][0], "FETCH_SIZE", 7.0),
                    (rows[0][0], "GRBM_GUI_ACTIVE", 8.0)])
    db.commit()
    db.close()
    out = pmc.read_db(str(tmp_path / "x_results.db"))
    assert out["k_trace_closest"]["calls"] == 2 and out["k_trace_closest"]["total_us"] == 8000.0
    assert out["k_trace_closest"]["FETCH_SIZE"] == 150.0 and out["k_trace_closest"]["GRBM_GUI_ACTIVE"] == 8.0
    assert out["k_shade"] == {"calls": 1, "total_us": 1000.0, "FETCH_SIZE": 7.0}


def test_every_pass_fits_the_counter_slots():
    """MI355X_MICROARCH.md "rocprofv3 PMC slots": 8 SQ counters per pass, FETCH_SIZE and WRITE_SIZE in separate passes."""
    for name, counters in pmc.PASSES.items():
        assert sum(c.startswith("SQ_") for c in counters) <= 8, name
        assert not ({"FETCH_SIZE", "WRITE_SIZE"} <= set(counters)), name
        assert sum(c.startswith("GRBM_") for c in counters) <= 2, name


def test_measure_fails_soft_without_a_gpu(tmp_path, monkeypatch):
    monkeypatch.setattr(pmc.shutil, "which", lambda name: None)
    monkeypatch.setattr(pmc.os.path, "exists", lambda p: False)
    res = pmc.measure(str(tmp_path / "none.bin"), str(tmp_path / "none.json"), passes=("fetch",))
    assert "error" in res["fetch"]
