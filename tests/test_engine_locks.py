"""The engine's reader / writer / lane gate can never make a caller wait for itself (round-3 review): more tickets than
lanes from one thread, lanes exhausted between threads, a writer waiting while a ticket holder asks for another ticket, a
writer call from a ticket holder.  Runs tests/c/locks_test.c against the product library -- the gate alone, no device."""
import subprocess

import qpelib as q


def test_lane_and_writer_gate_never_waits_for_its_caller(tmp_path):
    exe = tmp_path / "locks_test"
    subprocess.run(["gcc", "-std=c11", "-O2", "-Wall", "-Wextra", "-Werror", f"-I{q.ROOT / 'include'}", str(q.ROOT / "tests" / "c" / "locks_test.c"),
                    "-o", str(exe), f"-L{q.PKG}", "-lpqps_hip", "-lpthread", f"-Wl,-rpath,{q.PKG}"], check=True)
    r = subprocess.run([str(exe)], capture_output=True, text=True, timeout=120)
    assert r.returncode == 0, r.stdout + r.stderr
    assert "all ok" in r.stdout and "FAIL" not in r.stdout
    # the refusals say why, on stderr, like every diagnostic of the engine
    assert "already holds all 4 query lanes" in r.stderr
    assert "no query lane came free within 300 ms" in r.stderr
    assert "refused while the calling thread holds a query ticket" in r.stderr
