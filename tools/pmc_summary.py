"""Per-kernel averages of the PMC counters in a rocprofv3 rocpd database (small text for profiles/)."""
import sqlite3
import sys


def main(db, out, note=""):
    c = sqlite3.connect(db)
    rows = c.execute("select kernel_name, counter_name, avg(value), count(*) from counters_collection "
                     "where kernel_name like '%uhc_%' group by kernel_name, counter_name order by kernel_name, counter_name").fetchall()
    lines = [f"# rocprofv3 --pmc summary of {db}", f"# {note}", "# kernel | counter | mean value per dispatch | dispatches"]
    for k, n, v, cnt in rows:
        lines.append(f"{k[:90]} | {n} | {v:.6g} | {cnt}")
    open(out, "w").write("\n".join(lines) + "\n")
    print("\n".join(lines))


if __name__ == "__main__":
    main(sys.argv[1], sys.argv[2], " ".join(sys.argv[3:]))
