"""The engine over several device shards in ONE process (PQPS_DEVICES=0,1,...: include/buildEngine-hip.h).

The rows are split by the reference's block partition (executeEngine-mpi.c:703-715), every shard filters its
range, scan-mode results are concatenated in shard order and index-mode results merged per probe by
(key asc, row desc) -- the answers must be those of the single-device engine, i.e. the reference's goldens.
On a one-GPU box the same card is listed several times (each shard still has its own context, stream, scratch
and buffers); with two or more cards the first two are used.

The engine reads PQPS_DEVICES when it is created and the suite's engines are cached per process, so the
engine-level tests of the suite are re-run in one child process per shard layout."""
import os
import subprocess
import sys

import pytest

import qpelib as q

pq = q.pq
pytestmark = pytest.mark.gpu

ENGINE_TESTS = ("engine_select_matches_reference_golden or engine_boolprobe or kat_duplicates_and_ranges or kat_reference_unit_tests or "
                "linear_search_records or print_table_text or insert_then_delete or columnar_select or "
                "concurrent_callers or qpehip_prints")


def layouts():
    two_cards = pq.lib().pqps_device_count() >= 2
    return ["0,1" if two_cards else "0,0", "0,1,0" if two_cards else "0,0,0"]


def test_block_partition_and_answers_over_shards(tmp_path):
    """A direct look: the shard sizes are the mpi:703-715 partition; a scan, an index probe on each kind of
    key, a COUNT and a > 32-comparison list answer as the single-shard engine does."""
    code = r"""
import sys
sys.path.insert(0, %r)
import qpelib as q
pq = q.pq
csv = q.GOLDEN / "commands_2k.csv"
chains = [
    [("risk_level", ">", "3")],
    [("sudo_used", "=", "TRUE"), "AND", ("user_id", ">=", "1040")],
    [("command_id", ">=", "500"), "AND", ("command_id", "<", "700"), "OR", ("risk_level", "=", "5")],
    [("user_id", "<=", "1010"), "OR", ("exit_code", "!=", "0")],
    [],
    [("risk_level", ">", "9")],
]
wide = []
for i in range(40):
    wide += [("command_id", "=", str(17 * i + 3)), "OR"]
wide += [("user_id", "=", "1001")]
chains.append(wide)
eng = pq.HipEngine(csv, pq.DEFAULT_INDEXES)
print("SHARDS", eng.shards())
for chain in chains:
    print("IDS", eng.select_ids(chain))
    print("COUNT", eng.count(chain))
    cols = ["command_id", "user_name", "risk_level"]
    res = eng.select_columnar(cols, chain)
    print("COLUMNAR", res["rows"][:50], res["numRecords"])
    eng.free_columnar(res)
eng.close()
""" % str(q.ROOT / "tests")

    def run(devices):
        env = dict(os.environ)
        env.pop("PQPS_DEVICES", None)
        if devices:
            env["PQPS_DEVICES"] = devices
        p = subprocess.run([sys.executable, "-c", code], capture_output=True, text=True, timeout=600, env=env)
        assert p.returncode == 0, (p.stdout[-2000:], p.stderr[-2000:])
        return p.stdout.splitlines()

    one = run(None)
    assert one[0] == "SHARDS [2000]"
    for devices, sizes in zip(layouts(), ([1000, 1000], [667, 667, 666])):
        many = run(devices)
        assert many[0] == f"SHARDS {sizes}"
        assert many[1:] == one[1:], devices


@pytest.mark.parametrize("layout", [0, 1])
def test_engine_suite_over_shards(layout):
    devices = layouts()[layout]
    env = dict(os.environ, PQPS_DEVICES=devices)
    p = subprocess.run([sys.executable, "-m", "pytest", "-x", "-q", "-m", "gpu", "-p", "no:cacheprovider",
                        str(q.ROOT / "tests" / "test_gpu_parity.py"), str(q.ROOT / "tests" / "test_gpu_driver.py"),
                        "-k", ENGINE_TESTS],
                       capture_output=True, text=True, timeout=1500, env=env, cwd=str(q.ROOT))
    assert p.returncode == 0, (devices, p.stdout[-3000:], p.stderr[-2000:])
    assert " passed" in p.stdout and "no tests ran" not in p.stdout
