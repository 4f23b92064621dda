"""Per-image-pair hot path as one object: DINOv2-S/14 patch descriptors for both images of every
pair (HIP ViT) + dense mutual-NN matching (HIP matcher), and the one-process-per-GPU sharding of a
pair list with a final gather of match counts (RCCL over xGMI when backend='nccl').

The reference runs this loop serially at batch 1 (eval_linemod_json.py:52-169); pairs are
independent, so here they are batched per GPU and sharded across GPUs with no data-path
collective.
"""
import torch

from .matcher import dense_match


def shard_range(n_items, rank, world):
    """Contiguous block partition of `n_items` work items; sizes differ by at most one."""
    base, rem = divmod(n_items, world)
    start = rank * base + min(rank, rem)
    return start, start + base + (1 if rank < rem else 0)


def gather_counts(local_counts, group=None):
    """All ranks learn every pair's match count, in global pair order (the single exchange step).
    local_counts: int32 tensor [n_local] on the rank's device (CPU tensors for the gloo tests)."""
    import torch.distributed as dist
    if not (dist.is_available() and dist.is_initialized()) or dist.get_world_size(group) == 1:
        return local_counts.clone()
    world = dist.get_world_size(group)
    n_local = torch.tensor([local_counts.numel()], dtype=torch.int64, device=local_counts.device)
    sizes = [torch.zeros_like(n_local) for _ in range(world)]
    dist.all_gather(sizes, n_local, group=group)
    sizes = [int(s) for s in sizes]
    cap = max(sizes)
    padded = torch.zeros(cap, dtype=local_counts.dtype, device=local_counts.device)
    padded[:local_counts.numel()] = local_counts
    out = [torch.empty_like(padded) for _ in range(world)]
    dist.all_gather(out, padded, group=group)
    return torch.cat([o[:s] for o, s in zip(out, sizes)])


class PairPipeline:
    """extract(img0), extract(img1) -> dense_match, for a batch of pairs resident on one GPU."""

    def __init__(self, model, chunk=64, thr=0.2, border_rm=2, temperature=0.1, streams=1):
        self.model = model
        self.chunk = chunk
        self.thr, self.border_rm, self.temperature = thr, border_rm, temperature
        self.n_streams = streams
        self._streams = None

    @torch.no_grad()
    def extract(self, images):
        """[B,3,H,W] -> x_norm_patchtokens [B, H/14*W/14, dim], processed `chunk` images at a time.
        With streams > 1 consecutive chunks run on different HIP streams (own scratch each), so the
        HBM-bound kernels and the tile-quantisation tails of one chunk overlap the MFMA phases of another."""
        starts = list(range(0, images.shape[0], self.chunk))
        B, _, H, W = images.shape
        p = self.model.patch_size
        # every chunk writes its normalised tokens straight into its slice of one buffer: no concatenation pass
        tokens = torch.empty(B, 1 + (H // p) * (W // p), self.model.embed_dim, device=images.device, dtype=torch.float32)
        if self.n_streams <= 1 or len(starts) == 1:
            for s in starts:
                self.model(images[s:s + self.chunk], is_training=True, out_norm=tokens[s:s + self.chunk])
            return tokens[:, 1:]
        main = torch.cuda.current_stream(images.device)
        if self._streams is None:
            self._streams = [torch.cuda.Stream(images.device) for _ in range(self.n_streams)]
        for st in self._streams:
            st.wait_stream(main)
        for k, s in enumerate(starts):
            with torch.cuda.stream(self._streams[k % self.n_streams]):
                self.model(images[s:s + self.chunk], is_training=True, out_norm=tokens[s:s + self.chunk])
        for st in self._streams:
            main.wait_stream(st)
        return tokens[:, 1:]

    @torch.no_grad()
    def __call__(self, img0, img1):
        assert img0.shape == img1.shape
        n, _, H, W = img0.shape
        p = self.model.patch_size
        f0, f1 = self.extract(img0), self.extract(img1)
        hw_c = (H // p, W // p)
        out = dense_match(f0, f1, hw_c, hw_c, (H, W), self.thr, self.border_rm, self.temperature)
        out["feat0"], out["feat1"] = f0, f1
        return out
