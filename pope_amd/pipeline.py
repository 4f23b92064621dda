"""Per-image-pair hot path as one object: DINOv2-S/14 patch descriptors for both images of every
pair (HIP ViT) + dense mutual-NN matching (HIP matcher), and the one-process-per-GPU sharding of a
pair list with a final gather of match counts (RCCL over xGMI when backend='nccl').

The reference runs this loop serially at batch 1 (eval_linemod_json.py:52-169); pairs are
independent, so here they are batched per GPU and sharded across GPUs with no data-path
collective.
"""
import json

import numpy as np
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


def load_pair_list(path):
    """Work list of an evaluation stream in the reference's walk order (eval_linemod_json.py:41-58: objects, then
    rotation bins, then the bin's pairs).  File: {"objects": [{"dir": ..., "bins": {"0": [[idx0, idx1], ...], ...}}]}
    (tests/golden/linemod_pairs.json holds the ids of the reference's data/pairs/LINEMOD-test.json).
    Returns an int64 array [n_pairs, 4] of (object, bin, idx0, idx1) rows; the row number is the global pair id."""
    with open(path) as f:
        doc = json.load(f)
    rows = []
    for o, obj in enumerate(doc["objects"]):
        for key, pairs in obj["bins"].items():
            rows.extend((o, int(key), int(a), int(b)) for a, b in pairs)
    return np.asarray(rows, dtype=np.int64).reshape(-1, 4)


def walk_pair_list(n_pairs, process_batch, batch=128, rank=0, world=1, group=None, device="cpu"):
    """The per-pair loop of the reference's drivers (eval_linemod_json.py:41-169) for a list of independent pairs on
    `world` GPUs: rank r owns the contiguous block shard_range(n_pairs, r, world) and walks it in batches of `batch`
    pairs (the last batch of a shard is ragged); `process_batch(lo, hi)` returns the int32 match counts of global
    pairs lo..hi-1; the single exchange is the gather of all counts at the end.  Returns (counts of ALL pairs in
    list order [n_pairs] on `device`, number of batches this rank ran)."""
    lo, hi = shard_range(n_pairs, rank, world)
    parts, n_batches = [], 0
    for s in range(lo, hi, batch):
        e = min(s + batch, hi)
        c = torch.as_tensor(process_batch(s, e), dtype=torch.int32)
        if c.numel() != e - s:
            raise ValueError(f"process_batch({s}, {e}) returned {c.numel()} counts")
        parts.append(c.to(device))
        n_batches += 1
    local = torch.cat(parts) if parts else torch.zeros(0, dtype=torch.int32, device=device)
    return gather_counts(local, group), n_batches


class PairPipeline:
    """extract(img0), extract(img1) -> dense_match, for a batch of pairs resident on one GPU.

    f16x3 range guard (deferred form): every ViT chunk gets its own device flag word; the words are read at the
    step's existing synchronisation point (the matcher's count readback) and a flagged chunk is re-run on the fp32
    MFMA in place, followed by a new match — same process, no relaunch (`model.on_overflow` = "raise" raises)."""

    def __init__(self, model, chunk=64, thr=0.2, border_rm=2, temperature=0.1, streams=1, want_conf=False,
                 match_precision=None):
        self.model = model
        self.chunk = chunk
        self.thr, self.border_rm, self.temperature = thr, border_rm, temperature
        self.n_streams = streams
        self.want_conf = want_conf  # publish conf_matrix [n, L, S] (the drop-in CoarseMatching always does)
        self.match_precision = match_precision  # arithmetic of the L x S x C contraction (None: matcher default)
        self._streams = None
        self.reruns = 0

    def _chunk(self, images, tokens, s, flag=None, precision=None):
        self.model(images[s:s + self.chunk], is_training=True, out_norm=tokens[s:s + self.chunk], range_flag=flag,
                   precision=precision)

    @torch.no_grad()
    def extract(self, images, flags=None):
        """[B,3,H,W] -> final-norm tokens [B, 1 + H/14*W/14, dim] (row 0 = CLS), `chunk` images per launch sequence.
        `flags`: int32 [ceil(B/chunk)] zeroed device tensor receiving the chunks' f16x3 range-guard words (None: each
        chunk checks its own flag, one host synchronisation per chunk).
        With streams > 1 consecutive chunks run on different HIP streams (own scratch each), so the
        HBM-bound kernels and the tile-quantisation tails of one chunk overlap the MFMA phases of another."""
        starts = list(range(0, images.shape[0], self.chunk))
        B, _, H, W = images.shape
        p = self.model.patch_size
        ntok = 1 + (H // p) * (W // p)
        # every chunk writes its normalised tokens straight into its slice of one buffer: no concatenation pass
        tokens = torch.empty(B, ntok, self.model.embed_dim, device=images.device, dtype=torch.float32)
        fl = (lambda k: None) if flags is None else (lambda k: flags[k:k + 1])
        if self.n_streams <= 1 or len(starts) == 1:
            for k, s in enumerate(starts):
                self._chunk(images, tokens, s, fl(k))
            return tokens
        main = torch.cuda.current_stream(images.device)
        # the lazily built caches (weight planes, pos/bias table) are filled on the MAIN stream before the fork: side
        # stream 1 must not read pointers whose contents side stream 0 is still writing
        with torch.cuda.device(images.device):
            self.model._weights()
            self.model._posb(H, W, ntok)
        if self._streams is None:
            self._streams = [torch.cuda.Stream(images.device) for _ in range(self.n_streams)]
        for st in self._streams:
            st.wait_stream(main)
        for k, s in enumerate(starts):
            with torch.cuda.stream(self._streams[k % self.n_streams]):
                self._chunk(images, tokens, s, fl(k))
        for st in self._streams:
            main.wait_stream(st)
        return tokens

    def _match(self, t0, t1, hw_c, hw_i):
        return dense_match(t0[:, 1:], t1[:, 1:], hw_c, hw_c, hw_i, self.thr, self.border_rm, self.temperature,
                           precision=self.match_precision, on_overflow=self.model.on_overflow, want_conf=self.want_conf)

    @torch.no_grad()
    def __call__(self, img0, img1):
        assert img0.shape == img1.shape
        n, _, H, W = img0.shape
        p = self.model.patch_size
        n_chunks = -(-n // self.chunk)
        flags = torch.zeros(2 * n_chunks, dtype=torch.int32, device=img0.device)
        t0, t1 = self.extract(img0, flags[:n_chunks]), self.extract(img1, flags[n_chunks:])
        hw_c = (H // p, W // p)
        out = self._match(t0, t1, hw_c, (H, W))   # synchronises (count readback)
        bits = flags.cpu()
        if int(bits.abs().sum()):
            self.model.range_overflow(int(np.bitwise_or.reduce(bits.numpy())))   # raises under "raise"
            for k in range(n_chunks):
                if int(bits[k]):
                    self._chunk(img0, t0, k * self.chunk, precision="f32")
                if int(bits[n_chunks + k]):
                    self._chunk(img1, t1, k * self.chunk, precision="f32")
            self.reruns += 1
            out = self._match(t0, t1, hw_c, (H, W))
        out["feat0"], out["feat1"] = t0[:, 1:], t1[:, 1:]
        out["cls0"], out["cls1"] = t0[:, 0], t1[:, 0]
        return out
