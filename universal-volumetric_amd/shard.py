"""shard.py — how the path shards across GPUs: contiguous blocks of whole texture segments per rank
(a segment = KTX2_BATCH_SIZE frames sharing codebooks and the P-frame chain), geometry frames with the same indices,
and ONE collective: an all_gather of 4 x int64 per rank {geometry frames, texture segments, layers in the last
segment, bytes written} from which rank 0 derives what scripts/Encoder.py:103-154 (check_total_frames) computes."""
import torch
import torch.distributed as dist


def plan(n_frames: int, batch: int, world: int, rank: int):
    """-> (first_frame, n_frames_local, first_segment, n_segments_local) for `rank`."""
    n_seg = (n_frames + batch - 1) // batch
    lo = n_seg * rank // world
    hi = n_seg * (rank + 1) // world
    f_lo = lo * batch
    f_hi = min(hi * batch, n_frames)
    return f_lo, max(0, f_hi - f_lo), lo, hi - lo


def gather_counts(n_geo: int, n_seg: int, last_layers: int, n_bytes: int, device=None):
    """all_gather of the per-rank manifest counters; returns the (world, 4) int64 table on every rank."""
    mine = torch.tensor([n_geo, n_seg, last_layers, n_bytes], dtype=torch.int64, device=device)
    if not (dist.is_available() and dist.is_initialized()) or dist.get_world_size() == 1:
        return mine[None, :].cpu()
    out = [torch.zeros_like(mine) for _ in range(dist.get_world_size())]
    dist.all_gather(out, mine)
    return torch.stack(out).cpu()


def totals(table, batch: int):
    """geometry frame count, texture segment count, texture frame count = (segments-1)*batch + layers(last) (Encoder.py:124-130)."""
    n_geo = int(table[:, 0].sum()); n_seg = int(table[:, 1].sum())
    nz = [int(r[2]) for r in table if int(r[1]) > 0]
    last = nz[-1] if nz else 0
    return n_geo, n_seg, (n_seg - 1) * batch + last if n_seg else 0, int(table[:, 3].sum())
