"""The N > 1 path on CPU: 2 gloo ranks run the REAL pipeline plumbing -- ``PipelineBase._shard`` / ``_sa_states`` /
``garment_features`` and ``imagdressing_amd.dist`` -- sharding the image batch and receiving the packed garment features in
ONE broadcast from rank 0 (the same code path RCCL takes on GPUs; only the tensors live elsewhere).

The only stand-in is the garment UNet's *forward* (HIP-only: the product path has no CPU fallback): a module with the
engine's surface (``cfg``, ``dtype``, ``attn_processors``, ``forward_nhwc``) that fills the real ``CacheAttnProcessor2_0``
caches with seeded tensors of the right shapes.  The two-rank run with the real HIP UNets on a GPU is
``tests/test_dist_2rank_gpu.py``."""
import os
import socket

import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp


def _free_port():
    s = socket.socket(); s.bind(("127.0.0.1", 0)); p = s.getsockname()[1]; s.close()
    return p


def _make_pipe():
    from imagdressing_amd import unet as E
    from imagdressing_amd.adapter.attention_processor import CacheAttnProcessor2_0
    from imagdressing_amd.dressing_sd.pipelines.IMAGDressing_v1_pipeline import IMAGDressing_v1
    from tests.harness import SMALL, hidden_size_of
    cfg = dict(E.SD15_CONFIG, **SMALL)
    boc = cfg["block_out_channels"]
    names = []
    for i in range(3):
        for j in range(2):
            for a in ("attn1", "attn2"):
                names.append(f"down_blocks.{i}.attentions.{j}.transformer_blocks.0.{a}.processor")
    for i in range(1, 4):
        for j in range(3):
            for a in ("attn1", "attn2"):
                names.append(f"up_blocks.{i}.attentions.{j}.transformer_blocks.0.{a}.processor")
    for a in ("attn1", "attn2"):
        names.append(f"mid_block.attentions.0.transformer_blocks.0.{a}.processor")

    class GarmentUNet:
        """engine surface; forward = seeded tensors into the real Cache processors (what the HIP forward leaves behind)"""
        dtype = torch.bfloat16
        device = torch.device("cpu")
        calls = 0

        def __init__(self):
            self.cfg = cfg
            self.attn_processors = {n: CacheAttnProcessor2_0() for n in names}

        def forward_nhwc(self, x, t, ehs):
            GarmentUNet.calls += 1
            assert x.shape[0] == 1 and ehs.shape[0] == 1, "the garment UNet runs at batch 1 (reference discards the null half, :476-480)"
            h, w = x.shape[1], x.shape[2]
            g = torch.Generator().manual_seed(5)
            for n, p in self.attn_processors.items():
                c = hidden_size_of(n, boc)
                lv = 3 if n.startswith("mid_block") else (int(n.split(".")[1]) if n.startswith("down") else 3 - int(n.split(".")[1]))
                p.cache["hidden_states"] = torch.randn(1, (h >> lv) * (w >> lv), c, generator=g).to(self.dtype)

    class DenoisingUNet:
        device = torch.device("cpu")
        dtype = torch.bfloat16
    pipe = IMAGDressing_v1(vae=None, reference_unet=GarmentUNet(), unet=DenoisingUNet(), tokenizer=None, text_encoder=None,
                           image_encoder=None, ImgProj=None, scheduler=None, safety_checker=None, feature_extractor=None)
    return pipe, GarmentUNet


def _worker(rank, world, port, q):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        from imagdressing_amd import dist as D
        pipe, G = _make_pipe()
        ref_lat = torch.zeros(1, 4, 16, 16)
        cloth = torch.zeros(2, 16, 64)                  # [null; cond] pair as ImgProj returns it; only [-1:] is used
        feats = pipe._sa_states(ref_lat, cloth, True)   # real method: rank 0 computes, one broadcast, everyone unpacks
        assert G.calls == (1 if rank == 0 else 0), "only rank 0 may run the garment UNet"
        single, G2 = _make_pipe()
        exp = single.garment_features(ref_lat, cloth)   # what a single process computes locally
        ok = set(feats) == {n for n in exp if n.endswith("attn1.processor")} and all(torch.equal(feats[n], exp[n]) for n in feats)
        lat = torch.arange(7 * 4 * 2 * 2, dtype=torch.float32).view(7, 4, 2, 2)
        mine = pipe._shard(lat, True)                    # real method
        gathered = [None] * world
        dist.all_gather_object(gathered, mine.tolist())
        # an unusable garment size raises on EVERY rank before the collective (nobody is left blocked in the broadcast)
        raised = False
        try:
            pipe._sa_states(torch.zeros(1, 4, 18, 16), cloth, True)
        except ValueError:
            raised = True
        q.put((ok, sum(gathered, []) == lat.tolist(), len(feats), raised))
    finally:
        dist.destroy_process_group()


@pytest.mark.timeout(120)
def test_two_rank_garment_broadcast_and_sharding():
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker, args=(r, 2, port, q)) for r in range(2)]
    for p in procs:
        p.start()
    res = [q.get(timeout=100) for _ in procs]
    for p in procs:
        p.join(30)
        assert p.exitcode == 0
    assert all(r[0] and r[1] and r[3] for r in res), res
    assert res[0][2] == 16          # the 16 attn1 layers of the SD1.5 topology


def test_feature_layout_matches_stride2_rounding():
    """Level sizes follow the stride-2 convs (ceil), and sizes the UNet cannot run are refused up front."""
    from imagdressing_amd import dist as D
    pipe, _ = _make_pipe()
    lay = dict(D.feature_layout(pipe.reference_unet, (96, 72)))       # BASELINE configs[4]: 768x576
    assert lay["down_blocks.0.attentions.0.transformer_blocks.0.attn1.processor"][1] == 6912
    assert lay["down_blocks.2.attentions.1.transformer_blocks.0.attn1.processor"][1] == 432
    assert lay["mid_block.attentions.0.transformer_blocks.0.attn1.processor"][1] == 108
    with pytest.raises(ValueError):
        D.feature_layout(pipe.reference_unet, (65, 64))


def test_shard_bounds_cover_the_batch_for_every_world_size():
    """BASELINE configs[3] / [4]: 64 and 32 images over 8 ranks, plus remainders and more ranks than images: the shards are
    contiguous, disjoint, ordered, cover [0, n) and differ in size by at most one."""
    from imagdressing_amd import dist as D
    assert [D.shard_bounds(64, r, 8) for r in range(8)] == [(8 * r, 8 * r + 8) for r in range(8)]
    assert [D.shard_bounds(32, r, 8) for r in range(8)] == [(4 * r, 4 * r + 4) for r in range(8)]
    for n in (1, 3, 7, 8, 9, 31, 32, 33, 63, 64, 65):
        for w in (1, 2, 3, 4, 8):
            b = [D.shard_bounds(n, r, w) for r in range(w)]
            assert b[0][0] == 0 and b[-1][1] == n
            assert all(b[i][1] == b[i + 1][0] for i in range(w - 1))
            sizes = [hi - lo for lo, hi in b]
            assert max(sizes) - min(sizes) <= 1 and sizes == sorted(sizes, reverse=True)


def _failing_worker(rank, world, port, q):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        pipe, G = _make_pipe()

        def boom(*a, **k):
            raise RuntimeError("garment pass failed on rank 0")
        pipe.garment_features = boom            # only rank 0 ever calls it
        try:
            pipe._sa_states(torch.zeros(1, 4, 16, 16), torch.zeros(2, 16, 64), True)
            q.put((rank, "no error"))
        except RuntimeError as e:
            q.put((rank, str(e)))
    finally:
        dist.destroy_process_group()


@pytest.mark.timeout(120)
def test_rank0_failure_raises_on_every_rank():
    """A failure of the garment pass on rank 0 travels in the status element of the ONE packed broadcast: every rank raises,
    none is left denoising garbage or blocked in a later collective (advisor finding, round 2)."""
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_failing_worker, args=(r, 2, port, q)) for r in range(2)]
    for p in procs:
        p.start()
    res = dict(q.get(timeout=100) for _ in procs)
    for p in procs:
        p.join(30)
        assert p.exitcode == 0
    assert "failed on rank 0" in res[0] and "rank 0 failed" in res[1], res
