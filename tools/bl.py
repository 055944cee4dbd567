"""One line per bench.py JSON line on stdin: renders/s, ms per step (window 1), per-step median, stage times. (A/B runs)"""
import json, sys
for line in sys.stdin:
    line = line.strip()
    if not line.startswith("{"):
        continue
    d = json.loads(line)
    c = d.get("config", {})
    sm = c.get("step_ms") or {}
    st = d.get("stages") or {}
    print(f"{d['value']:9.1f} /s  {d['ms_per_step']:.4f} ms  median {sm.get('median', float('nan')):.4f}  "
          f"views {c.get('views_per_step')} fused_loss {c.get('fused_loss')}  " + " ".join(f"{k}={v:.4f}" for k, v in st.items() if isinstance(v, (int, float))))
