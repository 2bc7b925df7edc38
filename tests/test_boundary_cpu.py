"""Host-side logic that needs no GPU: the state_dict contract, config flattening, weight packing
(BatchNorm folding), the nearest-upsample table, the C-ABI export list, and 'fails loudly' behaviour."""
import ctypes
import json
import os
import re

import numpy as np
import pytest
import torch

from oracle import sepreformer_oracle as orc
from sepreformer_amd import lib as L
from sepreformer_amd.config import SepConfig, VARIANTS, load_model_kwargs, variant_yaml
from sepreformer_amd.engine import nearest_index
from sepreformer_amd.model import Model
from sepreformer_amd.params import count_parameters
from sepreformer_amd.synth import synth_mixture, synth_state_dict

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.mark.parametrize("variant", ["SepReformer_Base_WSJ0", "SepReformer_Large_DM_WHAMR", "SepReformer_Large_DM_WHAM"])
def test_state_dict_contract(variant):
    """Same keys, shapes, dtypes AND order as the reference module tree (listing generated from the
    imported reference; SURVEY.md section 8b 'state_dict contract')."""
    with open(os.path.join(ROOT, "tests", "golden", "state_dict_keys.json")) as f:
        want = json.load(f)[variant]
    m = Model.from_config(VARIANTS[variant], init_seed=0)
    got = [[k, list(v.shape), str(v.dtype).replace("torch.", "")] for k, v in m.state_dict().items()]
    assert got == want
    assert m.num_stages == 4 and m.num_spks == 2


def test_parameter_counts():
    # SURVEY.md section 2.2 probe numbers
    assert count_parameters(VARIANTS["SepReformer_Base_WSJ0"]) == 14_691_584
    assert count_parameters(VARIANTS["SepReformer_Base_WSJ0"], include_aux=False) == 14_147_840
    assert count_parameters(VARIANTS["SepReformer_Large_DM_WHAMR"]) == 56_816_384
    assert count_parameters(VARIANTS["SepReformer_Large_DM_WHAM"]) == 61_022_976


@pytest.mark.parametrize("variant", [v for v in VARIANTS if not v.startswith("tiny")])
def test_yaml_surface(variant):
    kw = load_model_kwargs(variant_yaml(variant))
    cfg = SepConfig.from_model_kwargs(**kw, per_level_split=VARIANTS[variant].per_level_split)
    assert cfg == VARIANTS[variant]
    assert cfg.model_kwargs() == kw
    import importlib
    mod = importlib.import_module(f"models.{variant}.model")
    m = mod.Model(**kw)
    assert isinstance(m, torch.nn.Module) and m.cfg == cfg


def test_config_rejects_unsupported():
    kw = VARIANTS["SepReformer_Base_WSJ0"].model_kwargs()
    kw["module_audio_enc"]["bias"] = True
    with pytest.raises(ValueError):
        SepConfig.from_model_kwargs(**kw)


def test_frames_and_padding():
    c = VARIANTS["SepReformer_Base_WSJ0"]
    assert c.frames(32000) == 7997 and c.padded_frames(7997) == 8000
    assert c.padded_frames(8000) == 8000          # no pad when already a multiple (module.py:229-230)
    assert c.frames(73596) == 18396 and c.padded_frames(18396) == 18400


def test_checkpoint_roundtrip(tmp_path):
    """reference utils/util_engine.py:98-106 checkpoint layout loads with strict=False / True."""
    cfg = VARIANTS["tiny"]
    m = Model.from_config(cfg, init_seed=1).load_synthetic_(3)
    path = tmp_path / "epoch.0007.pth"
    torch.save({"epoch": 7, "model_state_dict": m.state_dict(), "optimizer_state_dict": {},
                "train_loss": 0.0, "valid_loss": 0.0}, path)
    m2 = Model.from_config(cfg, init_seed=2)
    ck = torch.load(path, map_location="cpu")
    missing, unexpected = m2.load_state_dict(ck["model_state_dict"], strict=False)
    assert not missing and not unexpected
    for (k, a), (_, b) in zip(m.state_dict().items(), m2.state_dict().items()):
        assert torch.equal(a, b), k


def test_default_init_matches_reference_families():
    m = Model.from_config(VARIANTS["tiny"], init_seed=0)
    sd = m.state_dict()
    ls = sd["separator.enc_stages.0.g_block_1.block.gcfn.Layer_scale.layer_scale"]
    assert torch.allclose(ls, torch.full_like(ls, 1e-5))          # network.py:8
    w = sd["separator.enc_stages.0.g_block_1.block.gcfn.net1.1.weight"]
    assert float(w.abs().max()) <= 1 / np.sqrt(w.shape[1]) + 1e-6
    assert torch.equal(sd["separator.enc_stages.0.downconv.BN.running_var"], torch.ones(64))
    assert sd["separator.enc_stages.0.downconv.BN.num_batches_tracked"].dtype == torch.long


def test_synthetic_weights_are_deterministic_and_o1():
    cfg = VARIANTS["tiny"]
    a, b = synth_state_dict(cfg, 0), synth_state_dict(cfg, 0)
    assert all(torch.equal(a[k], b[k]) for k in a)
    c = synth_state_dict(cfg, 1)
    assert not torch.equal(a["out_layer.end_conv1x1.0.weight"], c["out_layer.end_conv1x1.0.weight"])
    ls = a["separator.enc_stages.0.g_block_1.block.gcfn.Layer_scale.layer_scale"]
    assert float(ls.min()) >= 0.5 and float(ls.max()) <= 1.5
    x = synth_mixture(2, 800, seed=5)
    assert torch.equal(x, synth_mixture(2, 800, seed=5)) and x.dtype == torch.float32


def test_nearest_index_matches_torch_tables(golden):
    g = golden("blocks_tiny")
    for key in g:
        if key.startswith("nearest."):
            _, src, dst = key.split(".")
            assert np.array_equal(nearest_index(int(src), int(dst)), g[key]), key
    assert np.array_equal(nearest_index(8, 16), np.arange(16) >> 1)
    assert np.array_equal(nearest_index(8, 8), np.arange(8))


def test_batchnorm_folding_equals_oracle():
    """pack.py folds eval BatchNorm into linear2 (CLA) and into scale/shift (DownConv)."""
    from sepreformer_amd import pack
    cfg = VARIANTS["tiny"]
    sd = synth_state_dict(cfg, 0)
    p = "separator.enc_stages.0.l_block_1.block.cla"
    pk = pack.Packed()
    pack.pack_cla(pk, sd, p)
    w2, b2 = pk.keep[6], pk.keep[7]
    x = torch.randn(3, 50, cfg.feat)
    want = torch.nn.functional.batch_norm(
        torch.nn.functional.linear(x, sd[p + ".linear2.weight"], sd[p + ".linear2.bias"]).permute(0, 2, 1),
        sd[p + ".BN.running_mean"], sd[p + ".BN.running_var"], sd[p + ".BN.weight"], sd[p + ".BN.bias"],
        False, 0.1, 1e-5).permute(0, 2, 1)
    got = torch.nn.functional.linear(x, w2, b2)
    assert orc.agreement_db(got, want) > 120
    q = "separator.enc_stages.0.downconv"
    pk = pack.Packed()
    pack.pack_down(pk, sd, q)
    w, scale, shift = pk.keep
    x = torch.randn(2, 40, cfg.feat)
    conv = torch.nn.functional.conv1d(x.permute(0, 2, 1), sd[q + ".down_conv.weight"], None, stride=2, padding=2, groups=cfg.feat)
    got = torch.nn.functional.gelu(conv * scale[None, :, None] + shift[None, :, None]).permute(0, 2, 1)
    assert orc.agreement_db(got, orc.down_conv(sd, q, x)) > 120
    assert tuple(w.shape) == (cfg.down_kernel, cfg.feat)


def test_abi_exports_every_declared_symbol():
    """The C-ABI library loads and exports exactly what include/sepr.h declares (no compute here)."""
    hdr = open(os.path.join(ROOT, "include", "sepr.h")).read()
    declared = set(re.findall(r"\b(sepr_[a-z0-9_]+)\s*\(", hdr))
    assert declared == set(L.SIGNATURES), declared ^ set(L.SIGNATURES)
    lib = L.load()
    for name in declared:
        assert hasattr(lib, name), name
    assert lib.sepr_version() == int(re.search(r"#define SEPR_VERSION (\d+)", hdr).group(1)) == L.ABI_VERSION
    # the latched A/B switches: defaults, and a reload picks up the environment (what the knob fixture of conftest.py relies on)
    assert [lib.sepr_knob(i) for i in range(5)] == [1, 1, 1, 1, 1] and lib.sepr_knob(99) == 0
    assert b"gfx950" in lib.sepr_build_info()
    # argument validation happens before any HIP call, so it is checkable without a device
    assert lib.sepr_workspace_bytes(L.OP_GCFN, 0, 8, 0, 128, 256, 2) == 0
    n, T, F = 2, 1000, 128
    want = 2 * n * T * 4 + 3 * F * n * T * 4
    got = lib.sepr_workspace_bytes(L.OP_GCFN, n, T, 0, F, 256, 2)
    assert want <= got <= want + 8192
    assert lib.sepr_gcfn_fwd(None, None, 1, 8, 128, None, None, 0, None) == L.SEPR_EINVAL
    assert lib.sepr_prof_start(0, 10) == L.SEPR_EINVAL


def test_struct_layouts_match_header():
    hdr = open(os.path.join(ROOT, "include", "sepr.h")).read()
    for cname, cls in (("sepr_gcfn_w", L.GcfnW), ("sepr_cla_w", L.ClaW), ("sepr_mha_w", L.MhaW),
                       ("sepr_down_w", L.DownW), ("sepr_split_w", L.SplitW), ("sepr_fuse_w", L.FuseW),
                       ("sepr_out_w", L.OutW), ("sepr_x3_w", L.X3W)):
        body = re.search(r"typedef struct \{([^}]*)\} " + cname + ";", hdr).group(1)
        fields = re.findall(r"(?:const float\*|const void\*|sepr_x3_w)\s*(\w+);", body)
        assert fields == [f for f, _ in cls._fields_], cname
        n_x3 = len(re.findall(r"sepr_x3_w\s+\w+;", body))
        assert ctypes.sizeof(cls) == 8 * (len(fields) - n_x3) + 16 * n_x3
    assert [f for f, _ in L.EgaW._fields_] == ["attn", "gate_ln_g", "gate_ln_b", "gate_w", "gate_b", "pe_k", "maxlen", "x3_gate", "fused_gate_p", "pe_k_planes", "fused_qkv_p", "fused_out_p"]


def test_pack_glumlp_fused_layout():
    """Fused Linear -> GLU -> Linear packing (SpkSplit / OutputLayer, csrc gcfn_fused3_kernel MODE 1): decode the chunked
    up-projection fragments (gate rows carry -log2 e), the constants block and the per-128-column down-projection blocks with
    their k-slot order and tile-pair row interleave; the decoded matrices must reproduce ``w2 . GLU(w1 x + b1)``."""
    from sepreformer_amd.pack import pack_glumlp_fused
    g = torch.Generator().manual_seed(3)
    F, H, N = 128, 96, 256
    w1, b1 = torch.randn(2 * H, F, generator=g) * 0.1, torch.randn(2 * H, generator=g) * 0.1
    w2 = torch.randn(N, H, generator=g) * 0.1
    w1pb, w2p = pack_glumlp_fused(w1, b1, w2)
    nch, KS = H // 32, F // 32
    nfrag = 4 * KS * 2 * 64 * 8 * 2
    assert tuple(w1pb.shape) == (nch, nfrag + 4096) and tuple(w2p.shape) == (N // 128, nch, 8, 2, 64, 8)
    w1p = w1pb[:, :nfrag].contiguous().view(torch.bfloat16).view(nch, 4, KS, 2, 64, 8)
    cst = w1pb[:, nfrag:].contiguous().view(torch.float32).view(nch, 1024)
    W1 = torch.zeros(2 * H, F, dtype=torch.float64)
    B1 = torch.zeros(2 * H, dtype=torch.float64)
    w1s = w1p[:, :, :, 0].double() + w1p[:, :, :, 1].double()
    for c in range(nch):
        for j in range(2):
            B1[32 * c + 16 * j:][:16] = cst[c, j * 160:j * 160 + 16].double()
            B1[H + 32 * c + 16 * j:][:16] = cst[c, j * 160 + 16:j * 160 + 32].double()
        for t in range(4):
            base = (0 if t < 2 else H) + 32 * c + 16 * (t & 1)
            for ks in range(KS):
                for gq in range(4):
                    for i in range(16):
                        W1[base + i, 32 * ks + 8 * gq: 32 * ks + 8 * gq + 8] = w1s[c, t, ks, gq * 16 + i]
    W2 = torch.zeros(N, H, dtype=torch.float64)
    w2s = w2p[:, :, :, 0].double() + w2p[:, :, :, 1].double()        # [half, c, ft, lane, 8]
    for h in range(N // 128):
        for c in range(nch):
            for ft in range(8):
                for gq in range(4):
                    for i in range(16):
                        for e in range(8):
                            n = 4 * gq + e if e < 4 else 16 + 4 * gq + e - 4
                            W2[128 * h + 32 * (ft // 2) + 8 * (i // 4) + 4 * (ft % 2) + i % 4, 32 * c + n] = w2s[h, c, ft, gq * 16 + i, e]
    x = torch.randn(5, F, generator=g).double()
    hcalc = x @ W1.t() + B1
    y = (hcalc[:, :H] / (1.0 + torch.exp2(hcalc[:, H:]))) @ W2.t()          # value * rcp(1 + exp2(pre-scaled gate))
    href = x @ w1.double().t() + b1.double()
    yref = (href[:, :H] * torch.sigmoid(href[:, H:])) @ w2.double().t()
    assert float((y - yref).abs().max() / yref.abs().max()) < 1e-4


def test_pack_glumlp_fold_is_the_decoder_behind_the_second_projection():
    """Folded main head (csrc gcfn_fused3_kernel MODE 2, reference modules/module.py:252-256 + :278-283 with masking = False): the decoded
    k-slot fragments of ``W_fold = wdec^T . W2`` with ``b_fold``, overlap-added at stride 4, must equal ConvTranspose1d(Linear(g))."""
    from sepreformer_amd.pack import pack_glumlp_fold
    g = torch.Generator().manual_seed(11)
    N, H, K, Lf = 256, 256, 16, 9
    w2, b2 = torch.randn(N, H, generator=g) * 0.1, torch.randn(N, generator=g) * 0.1
    wdec = torch.randn(N, 1, K, generator=g) * 0.1
    w2p, bf = pack_glumlp_fold(w2, b2, wdec)
    nch = H // 32
    assert tuple(w2p.shape) == (nch, 1, 2, 64, 8) and w2p.dtype == torch.bfloat16 and tuple(bf.shape) == (K,)
    ws = w2p[:, 0, 0].double() + w2p[:, 0, 1].double()                       # [c, lane, 8]
    Wf = torch.zeros(K, H, dtype=torch.float64)
    for c in range(nch):
        for gq in range(4):
            for i in range(16):
                for e in range(8):
                    n = 4 * gq + e if e < 4 else 16 + 4 * gq + e - 4
                    Wf[i, 32 * c + n] = ws[c, gq * 16 + i, e]
    gated = torch.randn(Lf, H, generator=g).double()
    taps = gated @ Wf.t() + bf.double()                                      # [L, K]
    wav = torch.zeros((Lf - 1) * 4 + K, dtype=torch.float64)
    for l in range(Lf):
        wav[4 * l:4 * l + K] += taps[l]
    basis = gated @ w2.double().t() + b2.double()                            # [L, N]
    ref = torch.nn.functional.conv_transpose1d(basis.t()[None], wdec.double(), stride=4)[0, 0]
    assert float((wav - ref).abs().max() / ref.abs().max()) < 1e-4


def test_pack_x3_layout_and_split():
    """bf16x3 packing: fragment order and hi+lo reconstruction (error <= 2^-16 relative)."""
    from sepreformer_amd.pack import pack_x3
    g = torch.Generator().manual_seed(0)
    w = torch.randn(48, 96, generator=g)
    p = pack_x3(w)                                   # [tile, step, plane, g, i, 8]
    assert tuple(p.shape) == (3, 3, 2, 4, 16, 8) and p.dtype == torch.bfloat16 and p.is_contiguous()
    rec = (p[:, :, 0].float() + p[:, :, 1].float())  # [tile, step, g, i, 8]
    rec = rec.permute(0, 3, 1, 2, 4).reshape(48, 96) # tile,i | step,g,8
    assert float((rec - w).abs().max() / w.abs().max()) < 2.0 ** -16
    # lane g*16+i of (tile t, step s) holds w[16t+i][32s+8g : 32s+8g+8]
    t, s_, gg, i = 2, 1, 3, 5
    hi = p[t, s_, 0, gg, i].float()
    assert torch.allclose(hi, w[16 * t + i, 32 * s_ + 8 * gg: 32 * s_ + 8 * gg + 8].to(torch.bfloat16).float())
    with pytest.raises(ValueError):
        pack_x3(torch.zeros(20, 64))


def test_x3_layernorm_folding():
    """pack.Packed.x3 folds LayerNorm's affine: (xhat*g + b) . W^T + c == xhat . (W*g)^T + (c + W.b)."""
    from sepreformer_amd import pack
    g = torch.Generator().manual_seed(1)
    W, c = torch.randn(32, 64, generator=g), torch.randn(32, generator=g)
    gam, bet = torch.rand(64, generator=g) + 0.5, torch.randn(64, generator=g) * 0.1
    pk = pack.Packed("bf16x3")
    x3 = pk.x3(W, c, gam, bet)
    assert x3.wp and x3.bias
    wp, bias = pk.keep[-2], pk.keep[-1]
    Wf = (wp[:, :, 0].float() + wp[:, :, 1].float()).permute(0, 3, 1, 2, 4).reshape(32, 64)
    xhat = torch.randn(10, 64, generator=g)
    want = (xhat * gam + bet) @ W.t() + c
    got = xhat @ Wf.t() + bias
    assert orc.agreement_db(got, want) > 85
    assert not pack.Packed("fp32").x3(W, c).wp


def test_pack_gcfn_fused_layout():
    """Fused-GCFN packing: un-permute the fragment / k-slot order and run the block in fp64 with the
    reconstructed (gamma-folded) matrices; must equal the oracle's GCFN up to the bf16 hi+lo split (2^-17)."""
    from sepreformer_amd.pack import pack_gcfn_fused
    cfg = VARIANTS["tiny"]
    sd = synth_state_dict(cfg, 0)
    p = "separator.enc_stages.0.g_block_1.block.gcfn"
    F = cfg.feat
    w1pb, w2p = pack_gcfn_fused(sd[p + ".net1.1.weight"], sd[p + ".net1.1.bias"], sd[p + ".net1.0.weight"],
                                sd[p + ".net1.0.bias"], sd[p + ".net2.2.weight"], sd[p + ".depthwise.weight"],
                                sd[p + ".depthwise.bias"])
    nch, KS = 3 * F // 32, F // 32
    nfrag = 4 * KS * 2 * 64 * 8 * 2                                   # fragment bytes per chunk
    assert tuple(w1pb.shape) == (nch, nfrag + 4096) and w1pb.dtype == torch.uint8
    assert tuple(w2p.shape) == (nch, F // 16, 2, 64, 8)
    w1p = w1pb[:, :nfrag].contiguous().view(torch.bfloat16).view(nch, 4, KS, 2, 64, 8)
    cst = w1pb[:, nfrag:].contiguous().view(torch.float32).view(nch, 1024)
    b1f = torch.zeros(6 * F)
    dwt = torch.zeros(6 * F, 3)
    dwb = torch.zeros(6 * F)
    for c in range(nch):
        for j in range(2):
            blk = cst[c, j * 160:(j + 1) * 160].view(10, 16)
            v, g_ = 32 * c + 16 * j, 3 * F + 32 * c + 16 * j
            b1f[v:v + 16], b1f[g_:g_ + 16] = blk[0], blk[1]
            dwt[v:v + 16] = blk[2:5].t()
            dwt[g_:g_ + 16] = blk[5:8].t()
            dwb[v:v + 16], dwb[g_:g_ + 16] = blk[8], blk[9]
    # the gate half (rows 3F..6F) of the conv constants is stored multiplied by -log2(e)
    gs = torch.ones(6 * F, dtype=torch.float64)
    gs[3 * F:] = -1.4426950408889634
    assert torch.equal(dwt, (sd[p + ".depthwise.weight"][:, 0, :].double() * gs[:, None]).float())
    assert torch.equal(dwb, (sd[p + ".depthwise.bias"].double() * gs).float())
    W1 = torch.zeros(6 * F, F, dtype=torch.float64)
    W2 = torch.zeros(F, 3 * F, dtype=torch.float64)
    w1s = (w1p[:, :, :, 0].double() + w1p[:, :, :, 1].double())      # [c, t, ks, lane, 8]
    w2s = (w2p[:, :, 0].double() + w2p[:, :, 1].double())            # [c, ft, lane, 8]
    for c in range(nch):
        for t in range(4):
            base = (0 if t < 2 else 3 * F) + 32 * c + 16 * (t & 1)
            for ks in range(KS):
                for g in range(4):
                    for i in range(16):
                        W1[base + i, 32 * ks + 8 * g: 32 * ks + 8 * g + 8] = w1s[c, t, ks, g * 16 + i]
        for ft in range(F // 16):
            for g in range(4):
                for i in range(16):
                    for e in range(8):
                        n = 4 * g + e if e < 4 else 16 + 4 * g + e - 4
                        # fragment row i = 4q + r of tile ft is output channel 32*(ft//2) + 8q + 4*(ft%2) + r
                        W2[32 * (ft // 2) + 8 * (i // 4) + 4 * (ft % 2) + i % 4, 32 * c + n] = w2s[c, ft, g * 16 + i, e]
    x = torch.randn(2, 21, F, dtype=torch.float64)
    xn = (x - x.mean(-1, keepdim=True)) / torch.sqrt(x.var(-1, unbiased=False, keepdim=True) + 1e-5)
    h = xn @ W1.t() + b1f.double()
    h = torch.nn.functional.conv1d(h.permute(0, 2, 1), sd[p + ".depthwise.weight"].double(), sd[p + ".depthwise.bias"].double(),
                                   padding=1, groups=6 * F).permute(0, 2, 1)
    g_ = h[..., : 3 * F] * torch.sigmoid(h[..., 3 * F:])
    y = x + (g_ @ W2.t() + sd[p + ".net2.2.bias"].double()) * sd[p + ".Layer_scale.layer_scale"].double()
    assert orc.agreement_db(y.float(), orc.gcfn(sd, p, x.float())) > 95


def test_no_cpu_fallback():
    m = Model.from_config(VARIANTS["tiny"], init_seed=0).eval()
    with pytest.raises(RuntimeError, match="HIP"):
        m(torch.zeros(1, 400))
    m.train()
    with pytest.raises(RuntimeError, match="HIP"):      # the training path is HIP-only too
        m(torch.zeros(1, 400))


def test_product_never_imports_oracle():
    pkg = os.path.join(ROOT, "sepreformer_amd")
    for fn in os.listdir(pkg):
        if fn.endswith(".py"):
            src = open(os.path.join(pkg, fn)).read()
            assert "import oracle" not in src and "from oracle" not in src, fn


# ---- packed layouts of the other fused kernels: decode them and redo the block in fp64 -----------------------------------------
def _unfrag(fr):
    """``[R][KS][2][64][8]`` bf16 fragments -> fp64 ``[R*16, KS*32]`` (hi + lo), the inverse of pack._split_frag."""
    R, KS = fr.shape[0], fr.shape[1]
    w = (fr[:, :, 0].double() + fr[:, :, 1].double()).view(R, KS, 4, 16, 8)      # [t, ks, g, i, e]
    return w.permute(0, 3, 1, 2, 4).reshape(R * 16, KS * 32)


def _unkslot(w2p):
    """``[C][FT][2][64][8]`` k-slot-ordered fragments -> fp64 ``[FT*16, C*32]``."""
    C, FT = w2p.shape[0], w2p.shape[1]
    w = (w2p[:, :, 0].double() + w2p[:, :, 1].double()).view(C, FT, 4, 16, 8)    # [c, ft, g, i, e]
    out = torch.zeros(FT * 16, C * 32, dtype=torch.float64)
    for g in range(4):
        for e in range(8):
            n = 4 * g + e if e < 4 else 16 + 4 * g + e - 4
            out[:, n::32] = w[:, :, g, :, e].permute(1, 2, 0).reshape(FT * 16, C)
    return out


def _chunks(wp, ntile, KS):
    """byte tensor ``[nch, frag + 4096]`` -> (fragments ``[nch, ntile, KS, 2, 64, 8]`` bf16, constants ``[nch, 1024]`` fp32)."""
    nfrag = ntile * KS * 2 * 64 * 8 * 2
    assert wp.dtype == torch.uint8 and wp.shape[1] == nfrag + 4096
    fr = wp[:, :nfrag].contiguous().view(torch.bfloat16).view(wp.shape[0], ntile, KS, 2, 64, 8)
    return fr, wp[:, nfrag:].contiguous().view(torch.float32).view(wp.shape[0], 1024)


@pytest.fixture(scope="module")
def base_sd():
    cfg = VARIANTS["SepReformer_Base_WSJ0"]
    return cfg, synth_state_dict(cfg, 0)


def _ln0(x):
    return (x - x.mean(-1, keepdim=True)) / torch.sqrt(x.var(-1, unbiased=False, keepdim=True) + 1e-5)


def test_pack_cla_fused_layout(base_sd):
    from sepreformer_amd.pack import pack_cla_fused, BN_EPS
    cfg, sd = base_sd
    p, F = "separator.enc_stages.0.l_block_1.block.cla", cfg.feat
    s = sd[p + ".BN.weight"].double() / torch.sqrt(sd[p + ".BN.running_var"].double() + BN_EPS)
    w2 = (sd[p + ".linear2.weight"].double() * s[:, None]).float()
    b2 = ((sd[p + ".linear2.bias"].double() - sd[p + ".BN.running_mean"].double()) * s + sd[p + ".BN.bias"].double()).float()
    w1p, w2p, w3p = pack_cla_fused(sd[p + ".linear1.weight"], sd[p + ".linear1.bias"], sd[p + ".layer_norm.weight"],
                                   sd[p + ".layer_norm.bias"], w2, b2, sd[p + ".linear3.1.weight"])
    KS = F // 32
    f1, c1 = _chunks(w1p, 4, KS)
    f2, c2 = _chunks(w2p, 4, KS)
    assert f1.shape[0] == F // 32 and f2.shape[0] == 2 * F // 64 and tuple(w3p.shape) == (2 * F // 32, F // 16, 2, 64, 8)
    W1, B1 = torch.zeros(2 * F, F, dtype=torch.float64), torch.zeros(2 * F, dtype=torch.float64)
    for c in range(F // 32):
        rows = _unfrag(f1[c])                                               # tiles v0 v1 g0 g1
        for t, base in enumerate([32 * c, 32 * c + 16, F + 32 * c, F + 32 * c + 16]):
            W1[base:base + 16], B1[base:base + 16] = rows[16 * t:16 * t + 16], c1[c, 16 * t:16 * t + 16].double()
    W2, B2 = torch.zeros(2 * F, F, dtype=torch.float64), torch.zeros(2 * F, dtype=torch.float64)
    for c in range(2 * F // 64):
        W2[64 * c:64 * c + 64], B2[64 * c:64 * c + 64] = _unfrag(f2[c]), c2[c, :64].double()
    W3 = _unkslot(w3p)
    x = torch.randn(2, 150, F, dtype=torch.float64)
    u = _ln0(x) @ W1.t() + B1
    u = u[..., :F] * torch.sigmoid(u[..., F:])
    cconv = torch.nn.functional.conv1d(u.permute(0, 2, 1), sd[p + ".dw_conv_1d.weight"].double(), sd[p + ".dw_conv_1d.bias"].double(),
                                       padding=32, groups=F).permute(0, 2, 1)
    h = torch.nn.functional.gelu(cconv @ W2.t() + B2)
    y = x + (h @ W3.t() + sd[p + ".linear3.1.bias"].double()) * sd[p + ".Layer_scale.layer_scale"].double()
    assert orc.agreement_db(y.float(), orc.cla(sd, p, x.float())) > 95


def test_pack_spk_fused_layout(base_sd):
    from sepreformer_amd.pack import pack_spk_fused
    cfg, sd = base_sd
    p, F, H = "separator.dec_stages.0.spk_attn_1.self_attn", cfg.feat, cfg.heads
    wqkv = torch.cat([sd[f"{p}.linear_{n}.weight"] for n in "qkv"], 0)
    bqkv = torch.cat([sd[f"{p}.linear_{n}.bias"] for n in "qkv"], 0)
    w1p, w2p = pack_spk_fused(wqkv, bqkv, sd[p + ".layer_norm.weight"], sd[p + ".layer_norm.bias"], sd[p + ".linear_out.weight"])
    KS = F // 32
    f1, c1 = _chunks(w1p, 6, KS)
    assert f1.shape[0] == F // 32 and tuple(w2p.shape) == (F // 32, F // 16, 2, 64, 8)
    W, Bq = torch.zeros(3 * F, F, dtype=torch.float64), torch.zeros(3 * F, dtype=torch.float64)
    for c in range(F // 32):
        rows = _unfrag(f1[c])                                               # tiles q(2c) q(2c+1) k(2c) k(2c+1) v(2c) v(2c+1)
        for t in range(6):
            base = (t // 2) * F + 16 * (2 * c + (t & 1))
            W[base:base + 16], Bq[base:base + 16] = rows[16 * t:16 * t + 16], c1[c, 16 * t:16 * t + 16].double()
    Wo = _unkslot(w2p)
    x = torch.randn(5, 2, F, dtype=torch.float64)                           # [frames, speakers, F]
    qkv = _ln0(x) @ W.t() + Bq
    q, k, v = (t_.view(5, 2, H, F // H).permute(0, 2, 1, 3) for t_ in qkv.split(F, -1))       # [n, H, S, dk]
    att = torch.softmax(q @ k.transpose(-1, -2) / (F // H) ** 0.5, -1) @ v
    o = att.permute(0, 2, 1, 3).reshape(5, 2, F)
    y = x + (o @ Wo.t() + sd[p + ".linear_out.bias"].double()) * sd[p + ".Layer_scale.layer_scale"].double()
    want = x.float() + orc.mha(sd, p, x.float(), None, H)
    assert orc.agreement_db(y.float(), want) > 95


def test_pack_gate_fused_layout(base_sd):
    from sepreformer_amd.pack import pack_gate_fused
    cfg, sd = base_sd
    p, F = "separator.enc_stages.0.g_block_1.block.ega", cfg.feat
    wp = pack_gate_fused(sd[p + ".block.linear.1.weight"], sd[p + ".block.linear.1.bias"], sd[p + ".block.linear.0.weight"],
                         sd[p + ".block.linear.0.bias"])
    fr, cst = _chunks(wp, 4, F // 32)
    W = torch.cat([_unfrag(fr[c]) for c in range(F // 64)], 0)
    b = torch.cat([cst[c, :64].double() for c in range(F // 64)], 0)
    x = torch.randn(3, 40, F, dtype=torch.float64)
    got = _ln0(x) @ W.t() + b
    ln = torch.nn.functional.layer_norm(x, (F,), sd[p + ".block.linear.0.weight"].double(), sd[p + ".block.linear.0.bias"].double(), 1e-5)
    want = ln @ sd[p + ".block.linear.1.weight"].double().t() + sd[p + ".block.linear.1.bias"].double()
    assert orc.agreement_db(got.float(), want.float()) > 95


def test_weights_key_and_replica_walk():
    """Packed weights are caches keyed by a cheap identity of the weights (data_ptr + version sum): in-place updates and
    ``invalidate_packed`` change it, and the tensors are found by attribute walk so that the parameter-less replicas
    of ``torch.nn.parallel.replicate`` (reference engine.py:64 with several device ids) resolve them too."""
    import torch
    from sepreformer_amd.config import VARIANTS
    from sepreformer_amd.model import Model
    m = Model.from_config(VARIANTS["tiny"], init_seed=0)
    k0 = m._weights_key()
    assert m._weights_key() == k0
    with torch.no_grad():
        m.out_layer.end_conv1x1._modules["2"].bias.add_(1.0)
    k1 = m._weights_key()
    assert k1 != k0
    m.out_layer.end_conv1x1._modules["2"].bias.data.mul_(2.0)      # invisible to version counters ...
    assert m._weights_key() == k1
    m.invalidate_packed()                                          # ... hence the explicit switch
    assert m._weights_key() != k1
    m.load_state_dict(m.state_dict())
    assert m._weights_key() != k1
    names = list(m.state_dict().keys())
    flat = m._flat_tensors()
    assert len(flat) == len(names) and all(a is b for a, b in zip(flat, m.state_dict(keep_vars=True).values()))
    # emulate replicate(): shallow module copies, parameters re-attached as plain tensors
    mods = list(m.modules())
    idx = {mm: i for i, mm in enumerate(mods)}
    reps = [mm._replicate_for_data_parallel() for mm in mods]
    for i, mm in enumerate(mods):
        for key, ch in mm._modules.items():
            setattr(reps[i], key, reps[idx[ch]])
        for key, p in mm._parameters.items():
            setattr(reps[i], key, p.detach().clone())
        for key, b in mm._buffers.items():
            setattr(reps[i], key, b.clone())
    rep = reps[0]
    assert rep._is_replica_module() and len(list(rep.parameters())) == 0
    rflat = rep._flat_tensors()
    assert len(rflat) == len(flat) and all(torch.equal(a, b) for a, b in zip(rflat, flat))
    assert rep._origin[0]() is m and rep._uid == m._uid
    with pytest.raises(RuntimeError, match="HIP"):
        rep(torch.zeros(1, 400))                                   # still no CPU fallback


def _header_struct_fields(hdr: str, cname: str):
    """Field (kind, name) list of a struct in include/sepr.h; handles several declarators per statement and comments."""
    body = re.search(r"typedef struct \{((?:[^{}]|\{[^{}]*\})*)\} " + cname + ";", hdr, re.S).group(1)
    body = re.sub(r"/\*.*?\*/", "", body, flags=re.S)
    out = []
    for stmt in body.split(";"):
        stmt = " ".join(stmt.split())
        if not stmt:
            continue
        m = re.match(r"(const float\*|const void\*|const sepr_u64\*|float\*|int|sepr_lin|sepr_mha_tw|sepr_mha_grad)\s*(.*)", stmt)
        assert m, (cname, stmt)
        for decl in m.group(2).split(","):
            out.append((m.group(1), decl.strip()))
    return out


def test_train_struct_layouts_match_header():
    """ctypes mirrors of the training-path structs (lib.py) against include/sepr.h: same fields, same order, same size."""
    hdr = open(os.path.join(ROOT, "include", "sepr.h")).read()
    size = {"const float*": 8, "const void*": 8, "const sepr_u64*": 8, "float*": 8, "int": 4, "sepr_lin": 32,
            "sepr_mha_tw": ctypes.sizeof(L.MhaTW), "sepr_mha_grad": ctypes.sizeof(L.MhaGrad)}
    pairs = [("sepr_lin", L.Lin), ("sepr_gcfn_tw", L.GcfnTW), ("sepr_gcfn_grad", L.GcfnGrad), ("sepr_cla_tw", L.ClaTW),
             ("sepr_cla_grad", L.ClaGrad), ("sepr_mha_tw", L.MhaTW), ("sepr_mha_grad", L.MhaGrad), ("sepr_ega_tw", L.EgaTW),
             ("sepr_ega_grad", L.EgaGrad), ("sepr_down_tw", L.DownTW), ("sepr_down_grad", L.DownGrad), ("sepr_split_tw", L.SplitTW),
             ("sepr_split_grad", L.SplitGrad), ("sepr_fuse_tw", L.FuseTW), ("sepr_fuse_grad", L.FuseGrad), ("sepr_out_tw", L.OutTW),
             ("sepr_out_grad", L.OutGrad), ("sepr_front_tw", L.FrontTW), ("sepr_front_grad", L.FrontGrad)]
    for cname, cls in pairs:
        fields = _header_struct_fields(hdr, cname)
        assert [n for _, n in fields] == [f for f, _ in cls._fields_], cname
        raw = sum(size[k] for k, _ in fields)
        assert ctypes.sizeof(cls) == (raw + 7) // 8 * 8, (cname, ctypes.sizeof(cls), raw)
    assert ctypes.sizeof(L.Lin) == 32           # three pointers + the planes switch, padded


def test_train_sizing_entries_run_without_a_device():
    """sepr_train_ctx_bytes / sepr_train_ws_bytes replay each block's carving in dry mode: no launch, no device needed."""
    lib = L.load()
    for op in range(11):
        c = lib.sepr_train_ctx_bytes(op, 4, 1000, 250, 128, 256, 2, 8)
        w = lib.sepr_train_ws_bytes(op, 4, 1000, 250, 128, 256, 2, 8, 65 if op == L.TOP_CLA else 5)
        assert c > 0 and w > 0, op
    # a GCFN keeps the LayerNorm statistics, the 6F hidden tensor and the gated 3F tensor
    n, T, F = 4, 1000, 128
    want = n * T * (2 + 6 * F + 3 * F) * 4
    got = lib.sepr_train_ctx_bytes(L.TOP_GCFN, n, T, 0, F, 256, 2, 8)
    assert want <= got <= want + 4096
    # the fused pair keeps the statistics only (the backward recomputes the hidden tensor)
    got_f = lib.sepr_train_ctx_bytes(L.TOP_GCFN_FUSED, n, T, 0, F, 256, 2, 8)
    assert n * T * 2 * 4 <= got_f <= n * T * 2 * 4 + 4096
    assert lib.sepr_train_ctx_bytes(L.TOP_GCFN, 0, T, 0, F, 256, 2, 8) == 0
    assert lib.sepr_gcfn_bwd(None, None, None, 1, 8, 128, None, None, None, 0, None, 0, 0.0, 0, None) == L.SEPR_EINVAL


def test_captured_train_step_refuses_what_it_cannot_capture():
    """train_step.CapturedTrainStep: no CPU path (the product fails loudly without the device), and an optimizer whose step counter
    lives on the host cannot be replayed from a graph - both are refused before anything runs."""
    import torch
    from sepreformer_amd.model import Model
    from sepreformer_amd.train_step import CapturedTrainStep
    m = Model.from_config(VARIANTS["tiny"], init_seed=0)
    x = torch.zeros(1, 2000)
    opt = torch.optim.AdamW(m.parameters(), lr=1e-4)
    with pytest.raises(RuntimeError, match="HIP device"):
        CapturedTrainStep(m, lambda a, b, *t: a[0].sum(), opt, x, [x, x])
    assert m.dropout_salt is None and m.train_graphs in (False, True)


def test_flat_adamw_has_no_cpu_path():
    """optim.FlatAdamW (include/sepr.h sepr_adamw_step) refuses parameters that are not on the HIP device - like every other entry
    of the product path, there is no CPU fallback - and validates its hyper-parameters before touching the library."""
    import torch
    from sepreformer_amd.config import VARIANTS
    from sepreformer_amd.model import Model
    from sepreformer_amd.optim import FlatAdamW
    m = Model.from_config(VARIANTS["tiny"], init_seed=0)
    with pytest.raises(RuntimeError, match="no CPU path"):
        FlatAdamW(m, lr=1e-3)
    with pytest.raises(ValueError):
        FlatAdamW(m, lr=1e-3, betas=(1.0, 0.999))
    assert FlatAdamW.fused_clip is True and issubclass(FlatAdamW, torch.optim.Optimizer)


def test_isa_lint_guards_the_gfx950_packed_f32_fault():
    """tools/isa_lint.py: the shipped library contains no packed-f32 instruction whose low half selects the high
    dword of src1 / src2 - the form that is wrong on gfx950 beside bf16 MFMAs (tools/probe/pk_opsel.hip) - and the lint recognises it."""
    import importlib.util
    spec = importlib.util.spec_from_file_location("isa_lint", os.path.join(ROOT, "tools", "isa_lint.py"))
    lint = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(lint)
    bad = ["v_pk_mul_f32 v[2:3], v[2:3], v[6:7] op_sel:[0,1]", "v_pk_add_f32 v[8:9], v[12:13], v[12:13] op_sel:[0,1] op_sel_hi:[1,0]",
           "v_pk_fma_f32 v[0:1], v[2:3], v[4:5], v[6:7] op_sel:[0,0,1]", "v_pk_fma_f32 v[0:1], v[2:3], v[4:5], v[6:7] op_sel:[1,1,0]"]
    good = ["v_pk_mul_f32 v[2:3], v[2:3], v[6:7]", "v_pk_mul_f32 v[2:3], v[2:3], v[6:7] op_sel_hi:[0,1]", "v_pk_fma_f32 v[0:1], v[2:3], v[4:5], v[6:7] op_sel:[1,0,0]",
            "v_pk_fma_f32 v[0:1], v[2:3], v[4:5], v[6:7] op_sel_hi:[0,1,1]", "v_pk_add_f32 v[2:3], v[2:3], v[6:7] op_sel_hi:[1,0] neg_lo:[0,1] neg_hi:[0,1]"]
    for ins in bad:
        assert lint.PK_F32.search(ins) and (lint.BAD_SEL.search(ins) or lint.BAD_SEL2.search(ins)), ins
    for ins in good:
        assert lint.PK_F32.search(ins) and not (lint.BAD_SEL.search(ins) or lint.BAD_SEL2.search(ins)), ins
    n_pk, found = lint.lint(L.LIB_PATH)
    assert n_pk > 5000, n_pk           # the disassembly really saw the device code (the library holds ~21 000 packed-f32 instructions)
    assert not found, found[:3]


def test_hot_kernels_are_spill_free():
    """hipcc hoists the per-thread offsets of a persistent-tile kernel out of its tile loop and spills them; the kernels re-derive their
    thread index per tile (SEPR_GB_REDERIVE / SEPR_GF3_REDERIVE / SEPR_XW_REDERIVE).  A change that brings the spills back costs 6 % of the largest training
    kernel without failing any parity test - so the shipped library is checked with tools/kres.py: the dominant kernels hold no spilled registers, and the
    library as a whole at most a handful (4 at the end of round 6, 25 before)."""
    import importlib.util
    spec = importlib.util.spec_from_file_location("kres", os.path.join(ROOT, "tools", "kres.py"))
    kres = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(kres)
    res = kres.resources(L.LIB_PATH)
    assert len(res) > 200, len(res)                       # the notes really list the device kernels (~260)
    by = {r["name"]: r for r in res}

    def one(sub):
        hits = [r for n, r in by.items() if sub in n]
        assert len(hits) == 1, (sub, [h["name"][:80] for h in hits])
        return hits[0]

    for sub in ("gcfn_bwd_mid_kernel<1, 2, true>", "gcfn_bwd_mid_kernel<3, 2, false>", "gcfn_fused3_kernel<128, 2, 4, 0, false, false, 0>",
                "gcfn_fused3_kernel<128, 2, 4, 0, true, true, 0>", "gcfn_fused3_kernel<128, 2, 4, 0, true, false, 0>", "gcfn_fused3_kernel<256, 2, 4, 0, false, false, 0>",
                "gemm_tnd_kernel<false, false>", "gemm_x3_kernel<0, 8, 48>", "cla_tail_kernel<128, 2>"):
        r = one(sub)
        assert int(r["spill"]) == 0 and int(r["scratch"]) == 0, (sub, r)
    assert int(one("gcfn_bwd_mid_kernel<1, 2, true>")["vgpr"]) <= 168      # three workgroups per CU
    spilling = [r["name"][:70] for r in res if int(r["spill"]) > 0]
    assert len(spilling) <= 6, spilling
