"""The N > 1 path on CPU: 2 gloo ranks shard the image batch and receive the packed garment features in
ONE broadcast from rank 0 (the same code path RCCL takes on GPUs; only the tensors live elsewhere)."""
import os
import socket

import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp


def _free_port():
    s = socket.socket(); s.bind(("127.0.0.1", 0)); p = s.getsockname()[1]; s.close()
    return p


def _worker(rank, world, port, q):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        from imagdressing_amd import dist as D
        from imagdressing_amd import unet as E
        from tests.harness import SMALL

        class RefUnet:                      # stands in for the garment UNet: names + shapes only
            cfg = dict(E.SD15_CONFIG, **SMALL)
            dtype = torch.bfloat16
            attn_processors = {}
        boc = RefUnet.cfg["block_out_channels"]
        names = []
        for i in range(3):
            for j in range(2):
                for a in ("attn1", "attn2"):
                    names.append(f"down_blocks.{i}.attentions.{j}.transformer_blocks.0.{a}.processor")
        names.append("mid_block.attentions.0.transformer_blocks.0.attn1.processor")
        RefUnet.attn_processors = {n: None for n in names}

        class Pipe:
            reference_unet = RefUnet
            device = torch.device("cpu")
            calls = 0

            def garment_features(self, ref_latents, cloth):
                Pipe.calls += 1
                g = torch.Generator().manual_seed(5)
                return {n: torch.randn(s, generator=g).to(torch.bfloat16) for n, s in D.feature_layout(RefUnet, (16, 16))}
        pipe = Pipe()
        feats = D.garment_features_broadcast(pipe, torch.zeros(1, 4, 16, 16), torch.zeros(1, 16, 64))
        assert Pipe.calls == (1 if rank == 0 else 0), "only rank 0 may run the garment UNet"
        exp = Pipe().garment_features(None, None)
        ok = all(torch.equal(feats[n], exp[n]) for n in feats) and all(n.endswith("attn1.processor") for n in feats)
        lat = torch.arange(7 * 4).view(7, 4)
        mine = D.shard_rows(lat)
        gathered = [None] * world
        dist.all_gather_object(gathered, mine.tolist())
        if rank == 0:
            q.put((ok, sum(gathered, []) == lat.tolist(), len(feats)))
        else:
            q.put((ok, True, len(feats)))
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
    assert all(r[0] and r[1] for r in res), res
    assert res[0][2] == 7          # 6 + 1 attn1 layers of the stand-in
