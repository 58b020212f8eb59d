"""Summarise a rocprofv3 (rocpd sqlite) kernel trace into a small text file for profiles/."""
import sqlite3
import sys


def main(db, out, note=""):
    c = sqlite3.connect(db)
    lines = [f"# rocprofv3 --kernel-trace --stats summary of {db}", f"# {note}",
             "# name | calls | total_ms | avg_ms | pct | grid | wg | lds_bytes | vgpr | agpr | sgpr"]
    meta = {}
    for r in c.execute("select name, grid_x, workgroup_x, lds_size, vgpr_count, accum_vgpr_count, sgpr_count from kernels group by name"):
        meta[r[0]] = r[1:]
    for name, calls, total, avg, pct in c.execute("select name,total_calls,total_duration,average,percentage from top_kernels"):
        short = name if len(name) < 110 else name[:107] + "..."
        m = meta.get(name, ("?",) * 6)
        lines.append(f"{short} | {calls} | {total / 1e3:.1f} | {avg / 1e3:.2f} | {pct:.3f} | {m[0]} | {m[1]} | {m[2]} | {m[3]} | {m[4]} | {m[5]}")
    open(out, "w").write("\n".join(lines[:70]) + "\n")
    print("\n".join(lines[:8]))


if __name__ == "__main__":
    main(sys.argv[1], sys.argv[2], " ".join(sys.argv[3:]))
