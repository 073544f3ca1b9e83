"""Calibration producers: histogram builder pinned against the reference's own
ActivationModule.find_histogram (fixture F7), file formats, and the greedy optimiser's behaviour."""
import csv

import numpy as np
import torch

from helpers import load_kat


def test_find_histogram_bit_identical_to_reference():
    from teal_amd.calibrate import find_histogram
    k = load_kat("kat_hist_producer.npz")
    acts = torch.from_numpy(k["acts"])
    nb = int(k["num_bins"])
    c1, m1 = find_histogram(acts[:3], nb)
    c2, m2 = find_histogram(acts[3:] * 2.0 + 0.1, nb)
    assert np.array_equal(c1.numpy(), k["h1"]) and np.array_equal(m1.numpy(), k["h1_centers"])
    assert np.array_equal(c2.numpy(), k["h2"]) and np.array_equal(m2.numpy(), k["h2_centers"])
    assert c1.sum() == acts[:3].numel() and c1[0] > 0 and c1[-1] > 0  # outlier bins hold the clipped 1 %


def test_grab_histograms_writes_reference_format_and_thresholds_hit_target(tmp_path):
    from teal_amd.calibrate import grab_histograms
    from teal_amd.distribution import Distribution, threshold_for_sparsity
    from teal_amd.gpt_fast import generate as G
    m = G.build_synthetic_model("tiny-test", "cpu", torch.float32, seed=3, std=0.05)
    ids = torch.randint(0, 512, (2, 96), generator=torch.Generator().manual_seed(0))
    grab_histograms(m, ids, str(tmp_path), num_bins=2000)
    for i in range(2):
        for sub in ("mlp", "self_attn"):
            h = torch.load(tmp_path / "histograms" / f"layer-{i}" / sub / "histograms.pt", weights_only=True)
            assert set(h) == {"h1", "h1_centers", "h2", "h2_centers"}
            assert all(v.shape == (2000,) and v.dtype == torch.float32 for v in h.values())
        assert (tmp_path / "activations" / f"act_{i}.pt").exists()
    # the thresholds derived from these files drop ~the requested fraction of fresh activations
    d = Distribution(str(tmp_path / "histograms" / "layer-1" / "mlp"), "h1")
    tau = threshold_for_sparsity(d, 0.5)
    acts = []
    hk = m.layers[1].feed_forward.register_forward_pre_hook(lambda mod, a: acts.append(a[0].flatten()))
    m.setup_caches(1, 64)
    with torch.no_grad():
        m(torch.randint(0, 512, (1, 64), generator=torch.Generator().manual_seed(5), dtype=torch.int), torch.arange(64))
    hk.remove()
    kept = float((torch.cat(acts).abs() > tau).float().mean())
    assert 0.4 < kept < 0.6, kept
    # and the decode-path plugin reads them like the reference's files
    from teal_amd.monkeypatch import layer_thresholds
    th = layer_thresholds(1, str(tmp_path / "histograms"), {p: [0.5, 0.5] for p in G.PROJS})
    assert th["gate"] == th["up"] == tau and th["q"] == th["k"] == th["v"]


def test_greedy_optimiser_csv_and_monotonicity(tmp_path):
    from teal_amd.calibrate import effective_sparsity, grab_histograms, greedy_optimize, weights_from_config
    from teal_amd.gpt_fast import generate as G
    from teal_amd.utils import get_layer_greedy_sparsities
    m = G.build_synthetic_model("tiny-test", "cpu", torch.float32, seed=3, std=0.05)
    ids = torch.randint(0, 512, (1, 48), generator=torch.Generator().manual_seed(1))
    grab_histograms(m, ids, str(tmp_path), num_bins=500)
    greedy_optimize(m, str(tmp_path), target_sparsity=0.3, base_step_size=0.1)
    w = weights_from_config(m.config)
    assert w["k"] == 0.5 and w["gate"] == 2.0  # tiny-test: 2 of 4 heads are KV heads, intermediate = 2 x dim
    for i in range(2):
        rows = list(csv.DictReader(open(tmp_path / "lookup" / f"layer-{i}" / "results.csv")))
        assert list(rows[0].keys()) == ["Effective Sparsity", "Activation Error", "Baseline Error", "q", "k", "v", "o", "gate", "up", "down"]
        eff = [float(r["Effective Sparsity"]) for r in rows]
        assert all(b > a for a, b in zip(eff, eff[1:])) and eff[-1] >= 0.3
        last = {p: float(rows[-1][p]) for p in w}
        assert abs(effective_sparsity(last, w) - eff[-1]) < 1e-9
    # the table is consumable by the reference-shaped lookup reader
    sp = get_layer_greedy_sparsities([0.2, 0.2], str(tmp_path / "lookup"))
    assert set(sp) == {"q", "k", "v", "o", "gate", "up", "down"} and len(sp["q"]) == 2


def test_greedy_optimiser_reproduces_the_reference_driver(tmp_path):
    """F9 (oracle/gen_golden.py:gen_greedy_driver): the reference's own process_layer (teal/greedyopt.py:99-159) ran on
    this tiny seeded block with the reference's SparsifyFn / Distribution wiring; greedy_optimize_layer must take the same
    decisions step by step (which projection is raised, by how much) and write the same rows."""
    import json
    import os

    from helpers import GOLDEN
    from teal_amd.calibrate import WEIGHT_DICT, _prefill_tables, greedy_optimize_layer
    from teal_amd.gpt_fast.model import ModelArgs, Transformer
    fx = os.path.join(GOLDEN, "greedy_driver")
    meta = json.load(open(os.path.join(fx, "meta.json")))
    model = Transformer(ModelArgs(**meta["config"])).eval()
    model.load_state_dict(torch.load(os.path.join(fx, "model.pt"), weights_only=True))
    weights = WEIGHT_DICT[meta["model_type"]]
    for i, layer in enumerate(model.layers):
        acts = torch.load(os.path.join(fx, "activations", f"act_{i}.pt"), weights_only=True)
        fc, mask = _prefill_tables(model, acts.shape[1], acts.device)
        out_csv = tmp_path / f"layer-{i}.csv"
        final = greedy_optimize_layer(layer, i, acts, os.path.join(fx, "histograms"), str(out_csv), weights, fc, mask,
                                      target_sparsity=meta["target_sparsity"], base_step_size=meta["base_step_size"],
                                      last_fraction=meta["last_fraction"])
        got = list(csv.reader(open(out_csv)))
        want = list(csv.reader(open(os.path.join(fx, "lookup", f"layer-{i}", "results.csv"))))
        assert got[0] == want[0] and len(got) == len(want) > 50
        for g, w in zip(got[1:], want[1:]):
            assert [float(v) for v in g[3:]] == [float(v) for v in w[3:]], "per-projection sparsities of a step"
            assert float(g[0]) == float(w[0])                                   # effective sparsity: same arithmetic
            assert np.allclose([float(g[1]), float(g[2])], [float(w[1]), float(w[2])], rtol=1e-4, atol=1e-6)
        assert {k: float(v) for k, v in final.items()} == {k: float(v) for k, v in meta[f"final_{i}"].items()}
