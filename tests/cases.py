"""Rebuild the inputs of a golden case (stored in full, or regenerated from its seed with the
same calls ``oracle/make_golden.py`` made) and check the regenerated inputs' digests."""
import torch

from oracle.make_golden import attn_weights, lora_weights, proj_plus_sd, resampler_sd, spike_tokens
from oracle.seeds import digest, seeded


def hybrid_inputs(c):
    if "x" in c:
        return {k: c[k] for k in ("x", "ref", "wq", "wk", "wv", "wo", "bo", "wk_ref", "wv_ref", "lora")}
    s, C = c["seed"], c["C"]
    d = attn_weights(s, C, C)
    d["x"] = seeded(s + 20, c["B"], c["N"], C)
    d["ref"] = seeded(s + 21, 1, c["M"], C)
    d["x"], d["ref"] = spike_tokens(d["x"], d["ref"], c.get("spike"))
    d["wk_ref"] = seeded(s + 22, C, C, scale=C ** -0.5)
    d["wv_ref"] = seeded(s + 23, C, C, scale=C ** -0.5)
    d["lora"] = lora_weights(s, C, C, c["rank"]) if c["rank"] else None
    assert digest(d["x"]) == c["digests"]["x"] and digest(d["ref"]) == c["digests"]["ref"]
    assert digest(d["wq"]) == c["digests"]["wq"] and digest(d["wk_ref"]) == c["digests"]["wkr"]
    return d


def cache_inputs(c):
    if "x" in c:
        return {k: c.get(k) for k in ("x", "ehs", "wq", "wk", "wv", "wo", "bo")}
    s, C = c["seed"], c["C"]
    d = attn_weights(s, C, c["KD"] or C)
    d["x"] = seeded(s + 20, c["B"], c["N"], C)
    d["ehs"] = seeded(s + 21, c["B"], c["T"], c["KD"], scale=0.5) if c["T"] else None
    assert digest(d["x"]) == c["digests"]["x"] and digest(d["wq"]) == c["digests"]["wq"]
    return d


def cross_inputs(c):
    if "x" in c:
        return {k: c[k] for k in ("x", "ehs", "wq", "wk", "wv", "wo", "bo", "wk_ip", "wv_ip", "lora")}
    s, C, KD = c["seed"], c["C"], c["KD"]
    d = attn_weights(s, C, KD)
    d["x"] = seeded(s + 20, c["B"], c["N"], C)
    d["ehs"] = seeded(s + 21, c["B"], c["T"] + c["ip_tokens"], KD, scale=0.5)
    d["lora"] = d["wk_ip"] = d["wv_ip"] = None
    if c["ip_tokens"]:
        d["lora"] = lora_weights(s, C, KD, c["rank"])
        d["wk_ip"] = seeded(s + 30, C, KD, scale=KD ** -0.5)
        d["wv_ip"] = seeded(s + 31, C, KD, scale=KD ** -0.5)
    assert digest(d["x"]) == c["digests"]["x"] and digest(d["ehs"]) == c["digests"]["ehs"]
    return d


def legacy_inputs(c):
    """Inputs of a tests/golden/processors_legacy.pt case (oracle/make_golden.py::legacy_case)."""
    s, C = c["seed"], c["C"]
    kd = c["KD"] or C
    d = attn_weights(s, C, kd)
    d["x"] = seeded(s + 20, c["B"], c["N"], C)
    d["ref"] = seeded(s + 21, 1, c["M"], C)
    d["ehs"] = seeded(s + 24, c["B"], c["T"], c["KD"], scale=0.5) if c["T"] else None
    d["wk_ref"] = d["wv_ref"] = None
    if c["kind"] == "refc":
        d["wk_ref"] = seeded(s + 22, C, C, scale=C ** -0.5)
        d["wv_ref"] = seeded(s + 23, C, C, scale=C ** -0.5)
    assert digest(d["x"]) == c["digests"]["x"] and digest(d["ref"]) == c["digests"]["ref"] and digest(d["wq"]) == c["digests"]["wq"]
    assert c["digests"]["ehs"] is None or digest(d["ehs"]) == c["digests"]["ehs"]
    return d


def resampler_inputs(c):
    if "x" in c:
        return c["sd"], c["x"]
    cfg = c["cfg"]
    sd = resampler_sd(c["seed"], cfg["dim"], cfg["depth"], cfg["dim_head"], cfg["heads"], cfg["num_queries"],
                      cfg["embedding_dim"], cfg["output_dim"])
    x = seeded(c["seed"] + 1000, c["B"], c["L"], cfg["embedding_dim"], scale=0.5)
    assert digest(x) == c["digests"]["x"] and digest(sd["latents"]) == c["digests"]["latents"]
    return sd, x


def proj_plus_inputs(c):
    s = c["seed"]
    sd = proj_plus_sd(s)
    idv = seeded(s + 2000, 1, 512)
    clip = seeded(s + 2001, 1, 257, 1280, scale=0.5)
    assert digest(idv) == c["digests"]["id"] and digest(clip) == c["digests"]["clip"]
    return sd, idv, clip


from tests.unet_fixture import unet_forward_inputs  # noqa: E402,F401  (re-export; the module itself imports no oracle code)
