#!/usr/bin/env python3
"""Writes N rows of the seeded synthetic commands_* table (the bench's table) as a CSV with the
reference's 12-column schema, for end-to-end runs of QPEHIP / QPESeq.
usage: scripts/make_csv.py N out.csv [seed]"""
import pathlib
import sys

ROOT = pathlib.Path(__file__).resolve().parent.parent
sys.path.insert(0, str(ROOT / "tests"))
import qpelib as q  # noqa: E402

pq = q.pq


def main():
    n, out = int(sys.argv[1]), sys.argv[2]
    seed = int(sys.argv[3], 0) if len(sys.argv) > 3 else 0x5EED
    h = q.HostSynth(n, seed=seed)
    a = h.arr
    shells = [s.decode() for s in pq.SYNTH_SHELLS]
    hosts = [s.decode() for s in pq.SYNTH_HOSTS]
    bases = [s.decode() for s in pq.SYNTH_BASES]
    with open(out, "w", newline="") as f:
        f.write("command_id,raw_command,base_command,shell_type,exit_code,timestamp,sudo_used,"
                "working_directory,user_id,user_name,host_name,risk_level\r\n")
        for i in range(n):
            uid = int(a["user_id"][i])
            base = bases[a["base_command"][i]]
            sudo = bool(a["sudo_used"][i])
            raw = ("sudo " if sudo else "") + base + " -x " + str(i % 97)
            ts = "2026-%02d-%02dT%02d:%02d:%02d.%03dZ" % (1 + i % 12, 1 + i % 28, i % 24, i % 60, (i * 7) % 60, i % 1000)
            f.write(f"{int(a['command_id'][i])},{raw},{base},{shells[a['shell_type'][i]]},{int(a['exit_code'][i])},{ts},"
                    f"{'true' if sudo else 'false'},/home/student{uid}/projects/cs{100 + i % 50},{uid},student{uid},"
                    f"{hosts[a['host_name'][i]]},{int(a['risk_level'][i])}\r\n")


if __name__ == "__main__":
    main()
