"""Summarise a rocprofv3 rocpd database: per-kernel calls / total / average / ms-per-step.  Usage: prof_summary.py db n_steps"""
import sqlite3
import sys

con = sqlite3.connect(sys.argv[1])
steps = int(sys.argv[2]) if len(sys.argv) > 2 else 1
rows = con.execute("select name,total_calls,total_duration,average,percentage from top_kernels").fetchall()
total = sum(r[2] for r in rows)
print(f"# total kernel time {total / 1e3:.1f} ms over {steps} steps = {total / 1e3 / steps:.2f} ms/step")
print("calls,total_us,avg_us,pct,ms_per_step,name")
for n, c, t, a, p in rows[:60]:
    short = n.replace("(anonymous namespace)::", "").replace("void ", "")[:120]
    print(f"{c},{t:.0f},{a:.2f},{p:.2f},{t / 1e3 / steps:.3f},{short}")
