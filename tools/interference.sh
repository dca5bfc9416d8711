#!/bin/bash
# diagnostic (GPU box): how much do one geometry context's serial kernels slow down next to (a) an HBM-streaming load, (b) an
# ALU/matrix load, (c) an LDS-hungry idle-ish load, each running in a second process on the same GPU?
run_bg() { python - "$1" <<'PY' &
import sys, time, torch
mode = sys.argv[1]; dev = torch.device("cuda:0"); t_end = time.time() + 55
if mode == "hbm":
    a = torch.empty(1 << 30, dtype=torch.uint8, device=dev); b = torch.empty_like(a)
    while time.time() < t_end:
        for _ in range(20): b.copy_(a)
        torch.cuda.synchronize()
elif mode == "alu":
    a = torch.randn(8192, 8192, device=dev, dtype=torch.bfloat16); b = torch.randn(8192, 8192, device=dev, dtype=torch.bfloat16)
    while time.time() < t_end:
        for _ in range(20): c = a @ b
        torch.cuda.synchronize()
elif mode == "valu":
    a = torch.randn(1 << 26, device=dev)
    while time.time() < t_end:
        for _ in range(20): a = torch.sin(a) * 1.0001 + 0.5
        torch.cuda.synchronize()
PY
}
for mode in none hbm alu valu; do
  if [ $mode != none ]; then run_bg $mode; BG=$!; sleep 12; fi
  EXTRA="" tools/sweep.sh if_$mode 240:1:geo:-:3
  if [ $mode != none ]; then wait $BG; fi
done
