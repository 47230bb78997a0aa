"""Print the device timeline (start, duration, gap to the previous end, grid) of the last N kernel dispatches / copies in a
rocprofv3 rocpd database (rocprofv3 --kernel-trace [--memory-copy-trace] -d DIR -o NAME -- ...  writes DIR/NAME_results.db).
    python tools/trace_timeline.py gpurun_out/prof_vr/vr_results.db [N] [anchor-substring]"""
import sqlite3, sys


def main():
    db = sqlite3.connect(sys.argv[1])
    n = int(sys.argv[2]) if len(sys.argv) > 2 else 40
    anchor = sys.argv[3] if len(sys.argv) > 3 else None
    rows = list(db.execute("select name,start,end,grid_x,grid_y,workgroup_x,workgroup_y from kernels order by start"))
    if anchor:
        idx = [i for i, r in enumerate(rows) if anchor in r[0]]
        rows = rows[idx[-3]:idx[-1] + 1] if len(idx) >= 3 else rows[-n:]
    else:
        rows = rows[-n:]
    t0, prev = rows[0][1], None
    busy = 0
    for name, s, e, gx, gy, wx, wy in rows:
        gap = (s - prev) / 1e3 if prev else 0.0
        busy += e - s
        print(f"{(s - t0) / 1e3:9.1f} dur {(e - s) / 1e3:7.1f} gap {gap:6.1f} grid {gx // max(wx, 1)}x{gy // max(wy, 1)} {name[:80]}")
        prev = max(prev or e, e)
    span = rows[-1][2] - t0
    print(f"span {span / 1e3:.1f} us, sum of durations {busy / 1e3:.1f} us")


if __name__ == "__main__":
    main()
