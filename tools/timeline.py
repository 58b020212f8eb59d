"""Per-step timeline of the step-kernel launches in a rocprofv3 (rocpd sqlite) kernel trace: for the last few control steps, start / end of
every uhc_step_kernel dispatch relative to the step's first launch (which tiers overlap, where the step waits).  Usage:
    python tools/timeline.py <results.db> <out.txt> [n_steps]"""
import sqlite3
import sys


def main(db, out, nsteps=3):
    c = sqlite3.connect(db)
    tabs = [r[0] for r in c.execute("select name from sqlite_master where type in ('table','view')")]
    kd = [t for t in tabs if "kernel_dispatch" in t][0] if any("kernel_dispatch" in t for t in tabs) else None
    rows = []
    if "kernels" in tabs:
        cols = [r[1] for r in c.execute("pragma table_info(kernels)")]
        st, en = ("start", "end") if "start" in cols else ("start_timestamp", "end_timestamp")
        gx = "grid_x" if "grid_x" in cols else "grid_size_x"
        rows = c.execute(f"select name, {st}, {en}, {gx}, lds_size from kernels order by {st}").fetchall()
    lines = [f"# step-kernel timeline from {db} (us relative to the first launch of the step; grid = workgroups, lds = bytes per workgroup)"]
    steps, cur = [], []
    for name, s, e, g, lds in rows:
        if "uhc_tier_lists_kernel" in name or ("uhc_step_kernel<0, 1" in name and not cur):
            if cur:
                steps.append(cur)
            cur = []
        if "uhc_step_kernel" in name or "uhc_step_queue_kernel" in name or "uhc_tier_lists" in name or "uhc_env_post" in name or "uhc_env_pre" in name:
            cur.append((name, s, e, g, lds))
    if cur:
        steps.append(cur)
    for k, stp in enumerate(steps[-nsteps:]):
        t0 = stp[0][1]
        lines.append(f"## step {len(steps) - nsteps + k}")
        for name, s, e, g, lds in stp:
            short = name.replace("void ", "").split("(")[0][:48]
            lines.append(f"{short:48s} start {1e-3 * (s - t0):9.1f}  end {1e-3 * (e - t0):9.1f}  dur {1e-3 * (e - s):9.1f}  grid {g:6d}  lds {lds}")
    open(out, "w").write("\n".join(lines) + "\n")
    print("\n".join(lines[-40:]))


if __name__ == "__main__":
    main(sys.argv[1], sys.argv[2], int(sys.argv[3]) if len(sys.argv) > 3 else 3)
