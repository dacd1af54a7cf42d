"""``inference_IMAGdressing*.py`` UNCHANGED: the reference scripts' own SOURCE TEXT is read from /root/reference, compiled (minus the
``if __name__ == "__main__"`` driver) and executed against ``<repo>/compat`` (the opt-in ``diffusers`` import shim) + ``<repo>``
(the ``adapter`` / ``dressing_sd`` shims) on ``sys.path``; then the script's own ``prepare(args)`` runs against synthetic local
checkpoints found through ``$IMD_MODEL_ROOT`` under the hub ids the script hard-codes (inference_IMAGdressing.py:42-52).

Runs in the build container only (no GPU needed: engines are constructed on the host and nothing is launched; the GPU box has
no /root/reference, where the restated flow of tests/test_e2e_gpu.py::test_prepare_flow_of_the_reference_script runs the same
sequence with a forward pass).  Out-of-scope host dependencies the scripts import but this image lacks (torchvision, cv2,
insightface, the tokenizer's vocabulary files) are stubbed HERE, in the test, not in the product."""
import ast
import json
import os
import sys
import types

import pytest
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
REF = "/root/reference"
pytestmark = pytest.mark.skipif(not os.path.isdir(REF), reason="the reference checkout exists in the build container only")


def _small_cfg():
    from imagdressing_amd import unet as E
    from tests.harness import SMALL
    return dict(E.SD15_CONFIG, **SMALL)


def _write_hf_dir(d, sd, cfg, fname="diffusion_pytorch_model.safetensors"):
    from safetensors.torch import save_file
    os.makedirs(d, exist_ok=True)
    save_file({k: v.contiguous() for k, v in sd.items()}, os.path.join(d, fname))
    with open(os.path.join(d, "config.json"), "w") as f:
        json.dump({k: (list(v) if isinstance(v, tuple) else v) for k, v in cfg.items()}, f)


@pytest.fixture()
def model_root(tmp_path, monkeypatch):
    """Synthetic 'downloads' laid out under the hub ids the scripts name."""
    from imagdressing_amd import unet as E
    from oracle import vae as OV
    from transformers import CLIPTextConfig, CLIPVisionConfig
    from transformers.models.clip.modeling_clip import CLIPTextModel as HFText          # (the library's own classes, whatever
    from transformers.models.clip.modeling_clip import CLIPVisionModelWithProjection as HFVision   # `transformers.X` points at)
    full = _small_cfg()
    keys = ("block_out_channels", "attention_head_dim", "norm_num_groups", "cross_attention_dim", "in_channels", "out_channels")
    sd_u = E.random_state_dict(E.unet_param_shapes(full), 0)
    _write_hf_dir(tmp_path / "SG161222" / "Realistic_Vision_V4.0_noVAE" / "unet", sd_u, {k: full[k] for k in keys})
    vcfg = dict(block_out_channels=(64, 128, 128, 128), norm_num_groups=8)
    _write_hf_dir(tmp_path / "stabilityai" / "sd-vae-ft-mse", OV.seeded_state_dict(vcfg, seed=0), vcfg)
    sd_c = E.random_state_dict(E.controlnet_param_shapes(full), 2, zero_convs=True)
    _write_hf_dir(tmp_path / "lllyasviel" / "control_v11p_sd15_openpose", sd_c, {k: full[k] for k in keys})
    _write_hf_dir(tmp_path / "lllyasviel" / "control_v11p_sd15_inpaint", sd_c, {k: full[k] for k in keys})       # ..._controlnetinpainting.py:148
    torch.manual_seed(0)
    tcfg = CLIPTextConfig(vocab_size=300, hidden_size=64, intermediate_size=128, num_hidden_layers=2, num_attention_heads=2,
                          max_position_embeddings=77, hidden_act="quick_gelu", projection_dim=64)
    HFText(tcfg).save_pretrained(tmp_path / "SG161222" / "Realistic_Vision_V4.0_noVAE" / "text_encoder", safe_serialization=True)
    vc = CLIPVisionConfig(hidden_size=96, intermediate_size=192, num_hidden_layers=2, num_attention_heads=2, image_size=28, patch_size=14,
                          projection_dim=64, hidden_act="gelu")
    HFVision(vc).save_pretrained(tmp_path / "h94" / "IP-Adapter" / "models" / "image_encoder", safe_serialization=True)
    monkeypatch.setenv("IMD_MODEL_ROOT", str(tmp_path))
    return tmp_path, full, sd_u


def _checkpoint(path, full, sd_u, emb_dim):
    """DeepSpeed-style IMAGDressing checkpoint (``{"module": {...}}``, key prefixes of inference_IMAGdressing.py:103-113)."""
    from imagdressing_amd import unet as E
    from imagdressing_amd.adapter import attention_processor as AP
    from imagdressing_amd.adapter.resampler import Resampler
    from tests.harness_names import attn_processor_names, hidden_size_of
    sd_r = E.random_state_dict(E.unet_param_shapes(full), 1)
    torch.manual_seed(3)
    cad = full["cross_attention_dim"]
    proj = Resampler(dim=cad, depth=4, dim_head=64, heads=12, num_queries=16, embedding_dim=emb_dim, output_dim=cad, ff_mult=4)   # the script's literals (:55-64)
    boc = full["block_out_channels"]
    procs = []
    for n in attn_processor_names(full):
        hs = hidden_size_of(n, boc)
        procs.append(AP.RefSAttnProcessor2_0(n, hs) if n.endswith("attn1.processor") else AP.CAttnProcessor2_0(n, hidden_size=hs, cross_attention_dim=cad))
    adapters = torch.nn.ModuleList(procs)
    with torch.no_grad():
        for p in adapters.parameters():
            p.copy_(torch.randn_like(p) * 0.05)
    ck = {}
    ck.update({"ref_unet." + k: v for k, v in sd_r.items()})
    ck.update({"unet." + k: v for k, v in sd_u.items()})
    ck.update({"proj." + k: v.detach().clone() for k, v in proj.state_dict().items()})
    ck.update({"adapter_modules." + k: v.detach().clone() for k, v in adapters.state_dict().items()})
    torch.save({"module": ck}, path)
    return ck


def _exec_script_defs(script, monkeypatch):
    """Execute the script's module body WITHOUT its ``if __name__ == "__main__"`` block; -> its namespace."""
    src = open(os.path.join(REF, script)).read()
    tree = ast.parse(src, filename=script)
    tree.body = [n for n in tree.body if not (isinstance(n, ast.If) and isinstance(n.test, ast.Compare)
                                              and getattr(n.test.left, "id", "") == "__name__")]
    # sys.path: the shims first -- what a user does with PYTHONPATH=<repo>/compat:<repo>
    monkeypatch.syspath_prepend(ROOT)
    monkeypatch.syspath_prepend(os.path.join(ROOT, "compat"))
    for m in [k for k in sys.modules if k == "diffusers" or k.startswith("diffusers.")]:
        monkeypatch.delitem(sys.modules, m)
    # host-side dependencies of the scripts that are outside the hot path and absent from this image: import-only stubs
    for name in ("torchvision", "torchvision.transforms", "cv2", "insightface", "insightface.app", "insightface.utils", "onnxruntime",
                 # the inpainting script's mask / pose preprocessing (SCHP human parsing, OpenPose: out of scope)
                 "preprocess", "preprocess.humanparsing", "preprocess.humanparsing.run_parsing", "preprocess.openpose",
                 "preprocess.openpose.run_openpose", "preprocess.utils_mask"):
        if name not in sys.modules:
            mod = types.ModuleType(name)
            mod.__dict__.update(transforms=None, FaceAnalysis=object, face_align=None, Parsing=object, OpenPose=object, get_mask_location=None)
            monkeypatch.setitem(sys.modules, name, mod)
    import transformers

    class FakeTokenizer:          # (no vocabulary files offline; the pipeline tests feed token ids / embeddings)
        model_max_length = 77

        @classmethod
        def from_pretrained(cls, *a, **k):
            return cls()
    monkeypatch.setattr(transformers, "CLIPTokenizer", FakeTokenizer, raising=False)
    # the shim re-points these two names at the engines when it is imported: have monkeypatch restore the library's own afterwards
    from transformers.models.clip import modeling_clip as _mc
    monkeypatch.setattr(transformers, "CLIPTextModel", _mc.CLIPTextModel, raising=False)
    monkeypatch.setattr(transformers, "CLIPVisionModelWithProjection", _mc.CLIPVisionModelWithProjection, raising=False)
    # host-only run: the VAE / CLIP constructors probe the device up front (no CPU path); here engines are only CONSTRUCTED
    # (weights repacked on the host), nothing is launched -- let construction through
    from imagdressing_amd import ops
    monkeypatch.setattr(ops, "ensure_device", lambda device: None)
    ns = {"__name__": "reference_script", "__file__": os.path.join(REF, script)}
    exec(compile(tree, os.path.join(REF, script), "exec"), ns)
    return ns


@pytest.mark.parametrize("script,pipeline_mod", [
    ("inference_IMAGdressing.py", "IMAGDressing_v1_pipeline"),
    ("inference_IMAGdressing_cartoon_style.py", "IMAGDressing_v1_pipeline"),
    ("inference_IMAGdressing_controlnetpose.py", "IMAGDressing_v1_pipeline_controlnet"),
    ("inference_IMAGdressing_ipa_controlnetpose.py", "IMAGDressing_v1_pipeline_ipa_controlnet"),
    ("inference_IMAGdressing_controlnetinpainting.py", "IMAGDressing_v1_pipeline_controlnet_inpainting"),      # the configs[4] surface
])
def test_reference_script_prepare_runs_unchanged(script, pipeline_mod, model_root, monkeypatch, tmp_path):
    root, full, sd_u = model_root
    ns = _exec_script_defs(script, monkeypatch)
    import diffusers
    assert diffusers.__file__.startswith(os.path.join(ROOT, "compat")), "the scripts' `from diffusers import ...` must hit the shim"
    from imagdressing_amd import clip as C
    from imagdressing_amd import unet as E
    from imagdressing_amd import vae as V
    assert ns["UNet2DConditionModel"] is E.UNet2DConditionModel and ns["AutoencoderKL"] is V.AutoencoderKL
    assert ns["CLIPTextModel"] is C.CLIPTextModel and ns["CLIPVisionModelWithProjection"] is C.CLIPVisionModelWithProjection   # IMD_NATIVE_CLIP default
    ck = _checkpoint(tmp_path / "IMAGDressing-v1_small.pt", full, sd_u, emb_dim=96)

    class args:
        device = "cpu"            # engines are CONSTRUCTED on the host here; every launch needs the GPU (tests/test_e2e_gpu.py)
        model_ckpt = str(tmp_path / "IMAGDressing-v1_small.pt")
    if "cartoon" in script:       # that script names a second base model and reads its VAE from the model ROOT (:42): same synthetic files
        tgt = root / "stablediffusionapi" / "counterfeit-v30"
        tgt.mkdir(parents=True)
        rv = root / "SG161222" / "Realistic_Vision_V4.0_noVAE"
        for sub in ("unet", "text_encoder"):
            os.symlink(rv / sub, tgt / sub)
        for f in os.listdir(root / "stabilityai" / "sd-vae-ft-mse"):
            os.symlink(root / "stabilityai" / "sd-vae-ft-mse" / f, tgt / f)
    if "ipa" in script:           # IP-Adapter FaceID-Plus checkpoint (image_proj + ip_adapter halves, ..._ipa_controlnet.py:88-101)
        from imagdressing_amd.adapter import attention_processor as AP0
        from imagdressing_amd.adapter.resampler import ProjPlusModel
        from tests.harness_names import attn_processor_names, hidden_size_of
        torch.manual_seed(5)
        cad = full["cross_attention_dim"]
        pp = ProjPlusModel(cross_attention_dim=cad, id_embeddings_dim=512, clip_embeddings_dim=96, num_tokens=4)
        ipl = torch.nn.ModuleList([AP0.LoraRefSAttnProcessor2_0(n, hidden_size_of(n, full["block_out_channels"])) if n.endswith("attn1.processor")
                                   else AP0.LoRAIPAttnProcessor2_0(hidden_size=hidden_size_of(n, full["block_out_channels"]), cross_attention_dim=cad,
                                                                   scale=1.0, rank=128, num_tokens=4) for n in attn_processor_names(full)])
        with torch.no_grad():
            for prm in ipl.parameters():
                prm.copy_(torch.randn_like(prm) * 0.05)
        ip_sd = {k: v.detach().clone() for k, v in ipl.state_dict().items() if "_ip" in k or "lora" in k}
        torch.save({"image_proj": pp.state_dict(), "ip_adapter": ip_sd}, tmp_path / "ip-adapter-faceid-plus_sd15.bin")
        args.ip_ckpt = str(tmp_path / "ip-adapter-faceid-plus_sd15.bin")
    pipe, generator = ns["prepare"](args)             # <- the reference's own function body, unmodified
    mod = __import__(f"imagdressing_amd.dressing_sd.pipelines.{pipeline_mod}", fromlist=["IMAGDressing_v1"])
    assert isinstance(pipe, mod.IMAGDressing_v1) and isinstance(generator, torch.Generator)
    assert isinstance(pipe.unet, E.UNet2DConditionModel) and isinstance(pipe.reference_unet, E.UNet2DConditionModel)
    assert isinstance(pipe.vae, V.AutoencoderKL) and isinstance(pipe.text_encoder, C.CLIPTextModel)
    from imagdressing_amd.adapter import attention_processor as AP
    procs = pipe.unet.attn_processors
    names = list(procs.keys())
    want = (AP.LoraRefSAttnProcessor2_0, AP.LoRAIPAttnProcessor2_0) if "ipa" in script else (AP.RefSAttnProcessor2_0, AP.CAttnProcessor2_0)
    assert all(isinstance(procs[n], want[0] if n.endswith("attn1.processor") else want[1]) for n in names)
    assert all(isinstance(p, AP.CacheAttnProcessor2_0) for p in pipe.reference_unet.attn_processors.values())
    # the adapter weights of the checkpoint arrived in the installed processors (index = position in unet.attn_processors, :86,:117)
    i0 = names.index("down_blocks.0.attentions.0.transformer_blocks.0.attn1.processor")
    got = procs[names[i0]].to_k_ref.weight.detach().float().cpu()
    assert torch.allclose(got, ck[f"adapter_modules.{i0}.to_k_ref.weight"].to(torch.float16).float(), atol=0)
    from imagdressing_amd.scheduler import DDIMScheduler
    assert isinstance(pipe.scheduler, DDIMScheduler)
    if "controlnet" in script:
        assert isinstance(pipe.controlnet, E.ControlNetModel)
    if "ipa" in script:           # the FaceID checkpoint reached the IP layers and the face projection model
        i1 = names.index("down_blocks.0.attentions.0.transformer_blocks.0.attn2.processor")
        assert torch.equal(procs[names[i1]].to_k_ip.weight.detach().float().cpu(), ip_sd[f"{i1}.to_k_ip.weight"].to(torch.float16).float())
        assert torch.equal(pipe.image_proj_model.state_dict()["norm.weight"].float().cpu(), pp.state_dict()["norm.weight"].float())


def test_stock_diffusers_unet_is_refused_with_directions():
    """Handing the pipelines anything that is a torch ``nn.Module`` without the engine surface -- what a stock diffusers
    ``UNet2DConditionModel`` is -- raises a TypeError that says what to build instead."""
    from imagdressing_amd.dressing_sd.pipelines.IMAGDressing_v1_pipeline import IMAGDressing_v1

    class UNet2DConditionModel(torch.nn.Module):          # stands in for diffusers.models.unet_2d_condition.UNet2DConditionModel
        def forward(self, sample, timestep, encoder_hidden_states):
            return (sample,)
    with pytest.raises(TypeError, match="imagdressing_amd.unet"):
        IMAGDressing_v1(vae=None, reference_unet=None, unet=UNet2DConditionModel(), tokenizer=None, text_encoder=None,
                        image_encoder=None, ImgProj=None, scheduler=None, safety_checker=None, feature_extractor=None)
