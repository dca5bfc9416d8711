"""tools/h2d_rate.py - what the host link of the GPU box delivers for copies out of page-locked memory (the PCIe-inclusive bench
variants are upload-bound): copy size swept, one and two streams.  Prints one JSON object."""
import json, time, torch

def rate(nbytes, reps, streams):
    src = [torch.empty(nbytes, dtype=torch.uint8).pin_memory() for _ in range(streams)]
    dst = [torch.empty(nbytes, dtype=torch.uint8, device="cuda") for _ in range(streams)]
    ss = [torch.cuda.Stream() for _ in range(streams)]
    for k in range(streams):
        with torch.cuda.stream(ss[k]):
            dst[k].copy_(src[k], non_blocking=True)
    torch.cuda.synchronize(); t = time.perf_counter()
    for _ in range(reps):
        for k in range(streams):
            with torch.cuda.stream(ss[k]):
                dst[k].copy_(src[k], non_blocking=True)
    torch.cuda.synchronize(); dt = time.perf_counter() - t
    return nbytes * reps * streams / dt / 1e9

out = {"what": "host -> device GB/s out of page-locked memory (torch copy_ non_blocking = hipMemcpyAsync), by copy size and streams"}
for nb, reps in ((1 << 20, 400), (4 << 20, 200), (16 << 20, 100), (128 << 20, 20)):
    for s in (1, 2):
        out["%d_MiB_x%d_streams" % (nb >> 20, s)] = round(rate(nb, reps, s), 1)
print(json.dumps(out))
