"""f64 flop (and HBM bytes) per env-step of a WHOLE workload from rocprofv3 --pmc passes: every step kernel of every tier added up.

    python tools/pmc_alu.py OUT.json ENV_STEPS VALU_DB [FETCH_DB WRITE_DB] [-- note]

VALU_DB: rocpd database of a pass with SQ_INSTS_VALU_{ADD,MUL,FMA,TRANS}_F64 and SQ_INSTS_VALU; FETCH_DB / WRITE_DB: passes with FETCH_SIZE /
WRITE_SIZE (KB, per the guide).  ENV_STEPS: env-steps the profiled command computed (n_env x control steps, warm-up included).  The
probes of bench.py spread their env-steps over three kernels that run side by side (fast / general / large tier), so their roofline is
taken per env-step of the workload, not per launch of one kernel (bench.py: alu_per_env_step)."""
import json
import sqlite3
import sys


def sums(db, like="%uhc_step%"):
    c = sqlite3.connect(db)
    rows = c.execute("select kernel_name, counter_name, sum(value), count(*) from counters_collection where kernel_name like ? group by kernel_name, counter_name", (like,)).fetchall()
    tot, per_kernel = {}, {}
    for k, n, v, cnt in rows:
        tot[n] = tot.get(n, 0.0) + v
        per_kernel.setdefault(k[:60], {})[n] = [v, cnt]
    return tot, per_kernel


def main():
    argv = sys.argv[1:]
    note = ""
    if "--" in argv:
        i = argv.index("--")
        note, argv = " ".join(argv[i + 1:]), argv[:i]
    out, env_steps, valu = argv[0], float(argv[1]), argv[2]
    t, pk = sums(valu)
    flop = 64.0 * (t.get("SQ_INSTS_VALU_ADD_F64", 0) + t.get("SQ_INSTS_VALU_MUL_F64", 0) + t.get("SQ_INSTS_VALU_TRANS_F64", 0) + 2.0 * t.get("SQ_INSTS_VALU_FMA_F64", 0))
    f64 = t.get("SQ_INSTS_VALU_ADD_F64", 0) + t.get("SQ_INSTS_VALU_MUL_F64", 0) + t.get("SQ_INSTS_VALU_TRANS_F64", 0) + t.get("SQ_INSTS_VALU_FMA_F64", 0)
    res = {"flop_per_env_step": flop / env_steps, "valu_wave_instructions_per_env_step": t.get("SQ_INSTS_VALU", 0) / env_steps,
           "f64_share_of_valu": (f64 / t["SQ_INSTS_VALU"]) if t.get("SQ_INSTS_VALU") else None, "env_steps": env_steps,
           "kernels": {k: {n: {"sum": v[0], "dispatches": v[1]} for n, v in d.items() if n in ("SQ_INSTS_VALU", "SQ_INSTS_VALU_FMA_F64")} for k, d in pk.items()},
           "formula": "64 x (ADD + MUL + TRANS + 2 FMA) f64 wave-instructions, summed over every dispatch of every uhc_step* kernel, / env-steps", "note": note}
    if len(argv) >= 5:
        f, _ = sums(argv[3])
        w, _ = sums(argv[4])
        res["hbm_bytes_per_env_step"] = 1024.0 * (f.get("FETCH_SIZE", 0) + w.get("WRITE_SIZE", 0)) / env_steps
        res["hbm_fetch_bytes_per_env_step"] = 1024.0 * f.get("FETCH_SIZE", 0) / env_steps
        res["hbm_write_bytes_per_env_step"] = 1024.0 * w.get("WRITE_SIZE", 0) / env_steps
    json.dump(res, open(out, "w"), indent=1)
    print(json.dumps({k: v for k, v in res.items() if k != "kernels"}))


if __name__ == "__main__":
    main()
