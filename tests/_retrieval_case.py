"""Synthetic retrieval case shared by tests/test_gpu_bf16.py and tests/probe_retrieval_modes.py: a handful of synthetic
"tracks", their log-mel segments as the reference cuts them (1 s windows, 50 % overlap), noisy copies as queries, a
briefly trained model, and fingerprints of the same segments from three generators (HIP bf16, HIP f32, CPU oracle)."""
import numpy as np
import torch


def synth_tracks(n, seconds, seed, dev, fs=16000):
    """n tracks (n, T) f32 on `dev` with slowly changing spectral content: five amplitude-modulated partials with random
    walks in frequency over a noise floor -- enough structure that 1 s segments are distinguishable while neighbouring
    segments stay similar.  The random knots come from a CPU generator (reproducible), the synthesis runs on the device."""
    import torch.nn.functional as F
    g = torch.Generator().manual_seed(seed)
    T = int(seconds * fs)
    x = 0.02 * torch.randn(n, T, generator=g).to(dev)
    for _ in range(5):
        f0 = 100.0 + 3500.0 * torch.rand(n, 1, generator=g)
        walk = f0 + torch.cumsum(torch.randn(n, T // 800 + 2, generator=g), 1) * 6.0
        env = torch.rand(n, T // 4000 + 2, generator=g)
        ph0 = 6.28 * torch.rand(n, 1, generator=g)
        f = F.interpolate(walk[:, None].to(dev, torch.float64), size=T, mode="linear", align_corners=True)[:, 0]
        e = F.interpolate(env[:, None].to(dev), size=T, mode="linear", align_corners=True)[:, 0]
        phase = (2 * np.pi / fs) * torch.cumsum(f, 1) + ph0.to(dev, torch.float64)     # float64: the phase must not drift
        x += 0.12 * e * torch.sin(phase).float()
    return x


def synth_track(seconds, seed, fs=16000):
    return synth_tracks(1, seconds, seed, torch.device("cpu"), fs)[0]


def add_noise(x, snr_db, seed):
    """x (..., T) + white noise at `snr_db` per row; the noise is drawn on x's device from a seeded generator."""
    g = torch.Generator(device=x.device).manual_seed(seed)
    n = torch.randn(x.shape, generator=g, device=x.device)
    rms = lambda a: a.pow(2).mean(dim=-1, keepdim=True).sqrt()
    return x + n * (rms(x) / rms(n)) * 10.0 ** (-snr_db / 20.0)


def segments(tracks, cfg, dev):
    """Log-mel segments of whole tracks on the device, as modules/transformations.py:89-90 cuts them -> (n, 64, 32)."""
    from grafp_amd import ops
    step = int(cfg["n_frames"] * (1 - cfg["overlap"]))
    out = []
    for x in tracks:
        spec = ops.logmel(x.to(dev), cfg["fs"], cfg["n_fft"], cfg["win_len"], cfg["hop_len"], cfg["n_mels"])
        out.append(ops.unfold_segments(spec, cfg["n_frames"], step))
    return torch.cat(out, dim=0)


def build_case(dev, n_tracks=24, seconds=20, train_steps=40, snrs=(0, 10), n_test=300, seed=0, overlap=None):
    from grafp_amd.train import Trainer, build_model
    from grafp_amd.util import load_config
    cfg = load_config()
    cfg["bsz_train"] = 64
    if overlap is not None:
        cfg["overlap"] = overlap          # hop between database segments (the reference's config: 0.9 = 0.1 s)
    torch.manual_seed(seed)
    model = build_model(cfg, device=dev)
    db_tracks = synth_tracks(n_tracks, seconds, 1000 + seed, dev)
    dummy_tracks = synth_tracks(n_tracks, seconds, 5000 + seed, dev)
    # a short training run on 1 s crops of the corpus (view j = the noisy crop): the net stops being a random projection
    tr = Trainer(cfg, model, dev, amp_dtype=torch.bfloat16, lr=2e-4)
    g = torch.Generator().manual_seed(seed + 1)
    pool = torch.cat([db_tracks, dummy_tracks])
    for it in range(train_steps):
        ti = torch.randint(0, pool.shape[0], (64,), generator=g)
        off = torch.randint(0, pool.shape[1] - 16000, (64,), generator=g)
        x_i = torch.stack([pool[a, b:b + 16000] for a, b in zip(ti.tolist(), off.tolist())])
        tr.step(x_i, add_noise(x_i, 5.0, 77 * it))
    model.eval()
    case = {"model": model, "cfg": cfg, "snrs": snrs, "lens": [1, 3, 5, 9],
            "db": segments(db_tracks, cfg, dev), "dummy": segments(dummy_tracks, cfg, dev)}
    for snr in snrs:
        case[f"query{snr}"] = segments(add_noise(db_tracks, float(snr), 900), cfg, dev)
    n_db = case["db"].shape[0]
    rng = np.random.RandomState(seed)
    case["test_ids"] = np.sort(rng.permutation(n_db - max(case["lens"]))[:n_test])
    return case


def fingerprints_hip(model, segs, dev, mode, batch=256):
    ctx = torch.autocast("cuda", dtype=torch.bfloat16) if mode == "bf16" else torch.autocast("cuda", enabled=False)
    out = []
    with torch.no_grad(), ctx:
        for i in range(0, segs.shape[0], batch):
            out.append(model.embed(segs[i:i + batch])[1].float().cpu())
    return torch.cat(out).numpy()


def fingerprints_oracle(model, segs, batch=64):
    """The CPU oracle's f32 forward on the same segments and weights (eval-mode BatchNorm)."""
    import torch.nn.functional as F

    from oracle import model as om
    torch.set_num_threads(max(1, min(32, torch.get_num_threads())))      # more threads only add contention here
    sd = {k: v.detach().cpu().clone() for k, v in model.state_dict().items()}
    x = segs.float().cpu()
    out = []
    with torch.no_grad():
        for i in range(0, x.shape[0], batch):
            h = om.graph_encoder(sd, om.peak_extract(sd, x[i:i + batch], 2), False)
            z = F.linear(F.elu(F.linear(h, sd["projector.0.weight"], sd["projector.0.bias"])),
                         sd["projector.2.weight"], sd["projector.2.bias"])
            out.append(F.normalize(z, p=2))
    return torch.cat(out).numpy()


def hit_rates(query, db, dummy, test_ids, lens):
    from oracle import retrieval
    return retrieval.eval_l2(query.astype(np.float32), db.astype(np.float32), dummy.astype(np.float32), test_ids, lens)[0]
