#!/bin/bash
# kernel durations (rocprofv3 kernel trace) of the prebuilt c3 kernel and of the same kernel compiled at creation
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
O=gpurun_out/jit_trace; rm -rf $O; mkdir -p $O
rocprofv3 --kernel-trace --stats --output-format csv -d $O/pre -- python3 bench.py --no-cpu-baseline --episodes synchronised --steps 1000 --warmup 200 --kernel-samples 0 > $O/pre.json 2> $O/pre.err
export MAPF_JIT_PREBUILT_TOO=1
rocprofv3 --kernel-trace --stats --output-format csv -d $O/jit -- python3 bench.py --no-cpu-baseline --episodes synchronised --steps 1000 --warmup 200 --kernel-samples 0 > $O/jit.json 2> $O/jit.err
python3 - <<'PY'
import pandas as pd, glob, numpy as np
for tag in ("pre", "jit"):
    f = sorted(glob.glob(f"gpurun_out/jit_trace/{tag}/**/*kernel_trace.csv", recursive=True))[-1]
    df = pd.read_csv(f); df = df[df.Kernel_Name.str.contains("k_step")].sort_values("Start_Timestamp")
    d = (df.End_Timestamp - df.Start_Timestamp).values
    gap = (df.Start_Timestamp.values[1:] - df.End_Timestamp.values[:-1])
    w = slice(300, 1200)
    print(tag, "kernel", df.Kernel_Name.iloc[0][:60], "median duration %.0f ns  median gap to next launch %.0f ns  median period %.0f ns" % (np.median(d[w]), np.median(gap[w]), np.median((df.Start_Timestamp.values[1:] - df.Start_Timestamp.values[:-1])[w])))
PY
