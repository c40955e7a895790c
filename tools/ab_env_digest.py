#!/usr/bin/env python3
"""Digest of tools/ab_env.sh's log: ms per step (and the kernels' ms per step) of every run.  usage: tools/ab_env_digest.py [log]"""
import json, sys
cfg = None
for line in open(sys.argv[1] if len(sys.argv) > 1 else "gpurun_out/ab_env.log"):
    if line.startswith("cfg "):
        cfg = line.strip()[4:]
    elif line.startswith("{"):
        d = json.loads(line)
        k = d.get("kernel_ms_per_step", {})
        print("%-60s %.3f ms per step   contraction %.3f   transport %.3f   check %s" % (cfg, d["ms_per_step"], k.get("k_jn_gemm", 0), k.get("k_transport", 0), d["check"]["ok"]))
