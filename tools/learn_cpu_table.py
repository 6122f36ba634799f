"""Table of the CPU-oracle learning control (tests/_learn_check_cpu.py runs, concatenated in profiles/r06_learning_cpu_control.txt
with one `== envs N gamma G` header per run) -> the markdown table of DESIGN section 7.
python tools/learn_cpu_table.py [profiles/r06_learning_cpu_control.txt]"""
import re
import sys

path = sys.argv[1] if len(sys.argv) > 1 else "profiles/r06_learning_cpu_control.txt"
runs, cur = [], None
for line in open(path):
    m = re.match(r"== envs (\d+) gamma ([\d.]+)", line)
    if m:
        cur = (int(m.group(1)), m.group(2), [])
        runs.append(cur)
        continue
    m = re.search(r"t=\s*(\d+)s\s+env-steps\s+(\d+)\s+updates\s+(\d+)\s+episodes\s+(\d+)\s+mean return\s+(-?[\d.]+)\s+mean len\s+([\d.]+)", line)
    if m and cur is not None:
        cur[2].append(tuple(float(x) for x in m.groups()))
print("| envs | γ | first 20 k-update window with mean return ≥ 200 (≥ 20 episodes) | at 100 k updates | at 200 k | at 400 k | last window |")
print("|---|---|---|---|---|---|---|")
for envs, g, pts in sorted(runs):
    first = next((p for p in pts if p[4] >= 200 and p[3] >= 20), None)

    def at(u):
        c = [p for p in pts if p[2] == u]
        return "%.0f (len %.0f)" % (c[0][4], c[0][5]) if c else "—"
    last = pts[-1] if pts else None
    print("| %d | %s | %s | %s | %s | %s | %s |" % (
        envs, g, ("%d k updates (%.0f s of one CPU core)" % (first[2] / 1000, first[0])) if first else "not reached", at(100000), at(200000), at(400000),
        ("%d k updates: %.0f (len %.0f)" % (last[2] / 1000, last[4], last[5])) if last else "—"))
