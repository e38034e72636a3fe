#!/usr/bin/env python3
"""Capture golden vectors from the REAL reference (runs only in the build container).

Imports ``/root/reference`` through the two shims of SURVEY.md Appendix A
(metadata version patch, tensorly stub), runs the hot-path estimators / losses
on small stored inputs and writes ``tests/golden/*.npz`` (inputs + outputs).
The reference source never travels; only these data files are committed.

    python tools/gen_golden.py
"""

from __future__ import annotations

import importlib.metadata as md
import os
import sys
import types

import numpy as np

sys.dont_write_bytecode = True
REF = "/root/reference"
if not os.path.isdir(REF):
    sys.exit("reference not mounted; goldens can only be regenerated in the build container")
sys.path.insert(0, REF)
_orig_version = md.version
md.version = lambda name: "0.0.0+oracle" if name == "cca_zoo" else _orig_version(name)
_tl = types.ModuleType("tensorly")
_tl.set_backend = lambda *a, **k: None
_dec = types.ModuleType("tensorly.decomposition")


def _nope(*a, **k):
    raise RuntimeError("tensorly stub")


_dec.parafac = _nope
_tl.decomposition = _dec
sys.modules["tensorly"] = _tl
sys.modules["tensorly.decomposition"] = _dec

import torch  # noqa: E402
from cca_zoo.datasets import JointData  # noqa: E402
from cca_zoo.deep.objectives import CCALoss, MCCALoss, _inv_sqrtm  # noqa: E402
from cca_zoo.linear import CCA, GCCA, MCCA, PLS, rCCA  # noqa: E402
from cca_zoo._utils._linalg import gevp, svd_whiten  # noqa: E402

OUT = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests", "golden")
os.makedirs(OUT, exist_ok=True)


def fitted(model, train, fresh, tag, store):
    model.fit(train)
    for i, w in enumerate(model.weights_):
        store[f"{tag}/w{i}"] = np.asarray(w)
    for i, mu in enumerate(model.means_):
        store[f"{tag}/mean{i}"] = np.asarray(mu)
    store[f"{tag}/score_train"] = model.score(train)
    store[f"{tag}/score_fresh"] = model.score(fresh)
    store[f"{tag}/pairwise_train"] = model.pairwise_correlations(train)
    for i, t in enumerate(model.transform(train)):
        store[f"{tag}/transform{i}"] = t[:5]
    for i, l in enumerate(model.get_factor_loadings(train)):
        store[f"{tag}/loadings{i}"] = l


def linear_case(name, train, fresh, specs):
    store = {}
    for i, v in enumerate(train):
        store[f"train{i}"] = v
    for i, v in enumerate(fresh):
        store[f"fresh{i}"] = v
    for tag, ctor in specs.items():
        fitted(ctor(), train, fresh, tag, store)
    np.savez_compressed(os.path.join(OUT, name + ".npz"), **store)
    print(name, len(store), "arrays")


# ---- C1: the README quick-start shape (README.md:56-64) --------------------
jd = JointData(n_views=2, n_samples=200, n_features=[50, 50], latent_dimensions=2,
               signal_to_noise=2.0, random_state=0)
c1_train, c1_fresh = jd.sample(), jd.sample()
two_view_specs = {
    "cca": lambda: CCA(latent_dimensions=2),
    "rcca_0.1": lambda: rCCA(latent_dimensions=2, c=0.1),
    "rcca_0.1_0.3": lambda: rCCA(latent_dimensions=2, c=[0.1, 0.3]),
    "pls": lambda: PLS(latent_dimensions=2),
    "rcca_0.1_nocenter": lambda: rCCA(latent_dimensions=2, c=0.1, center=False),
    "mcca_c0_pca": lambda: MCCA(latent_dimensions=2, c=0.0, pca=True),
    "mcca_c0_nopca": lambda: MCCA(latent_dimensions=2, c=0.0, pca=False),
    "mcca_c0.1_pca": lambda: MCCA(latent_dimensions=2, c=0.1, pca=True),
    "mcca_c0.1_nopca": lambda: MCCA(latent_dimensions=2, c=0.1, pca=False),
    "mcca_c0.1_nocenter": lambda: MCCA(latent_dimensions=2, c=0.1, center=False),
    "gcca_c0": lambda: GCCA(latent_dimensions=2, c=0.0),
    "gcca_c0.1": lambda: GCCA(latent_dimensions=2, c=0.1),
    "gcca_c0.1_nocenter": lambda: GCCA(latent_dimensions=2, c=0.1, center=False),
}
linear_case("c1_two_view_f64", c1_train, c1_fresh, two_view_specs)
# JointData stream itself (weights drawn first, then z, then noise per view)
np.savez_compressed(os.path.join(OUT, "jointdata_seed0.npz"),
                    draw0_v0=c1_train[0], draw0_v1=c1_train[1],
                    draw1_v0=c1_fresh[0], draw1_v1=c1_fresh[1])

# non-zero means + float32 inputs: rCCA stays fp32, MCCA/GCCA promote to fp64
off_train = [(v + 3.0 * (i + 1)).astype(np.float32) for i, v in enumerate(c1_train)]
off_fresh = [(v + 3.0 * (i + 1)).astype(np.float32) for i, v in enumerate(c1_fresh)]
linear_case("c1_two_view_f32_offset", off_train, off_fresh, {
    "rcca_0.1": lambda: rCCA(latent_dimensions=2, c=0.1),
    "cca": lambda: CCA(latent_dimensions=2),
    "mcca_c0.1": lambda: MCCA(latent_dimensions=2, c=0.1),
    "gcca_c0.1": lambda: GCCA(latent_dimensions=2, c=0.1),
})

# ---- three views, unequal widths, non-uniform view weights ------------------
jd3 = JointData(n_views=3, n_samples=300, n_features=[40, 30, 20], latent_dimensions=3,
                random_state=1)
t3, f3 = jd3.sample(), jd3.sample()
linear_case("three_view_f64", t3, f3, {
    "mcca_c0": lambda: MCCA(latent_dimensions=3, c=0.0),
    "mcca_c_list": lambda: MCCA(latent_dimensions=3, c=[0.1, 0.2, 0.3], pca=False),
    "gcca_c0": lambda: GCCA(latent_dimensions=3, c=0.0),
    "gcca_weighted": lambda: GCCA(latent_dimensions=3, c=0.1, view_weights=[1.0, 1.0, 2.0]),
    "gcca_nocenter": lambda: GCCA(latent_dimensions=3, c=0.2, center=False),
})

# ---- separated spectrum (well-posed per-column parity), k up to 6 -----------
rng = np.random.default_rng(7)
klat = 6
zs = rng.standard_normal((500, klat)) * np.linspace(2.0, 0.5, klat)
sep = [zs @ rng.standard_normal((klat, d)) + rng.standard_normal((500, d)) for d in (24, 17)]
zs2 = rng.standard_normal((500, klat)) * np.linspace(2.0, 0.5, klat)
sep_fresh = [zs2 @ rng.standard_normal((klat, d)) + rng.standard_normal((500, d)) for d in (24, 17)]
linear_case("separated_two_view_f64", sep, sep_fresh, {
    "cca_k6": lambda: CCA(latent_dimensions=6),
    "rcca_k6_c0.2": lambda: rCCA(latent_dimensions=6, c=0.2),
    "mcca_k6_c0.05": lambda: MCCA(latent_dimensions=6, c=0.05),
    "gcca_k6_c0.05": lambda: GCCA(latent_dimensions=6, c=0.05),
    "cca_k40": lambda: CCA(latent_dimensions=40),          # k clamps to min(d1, d2) = 17
})

# ---- rank-deficient views (d > n) with c > 0: well posed --------------------
rng = np.random.default_rng(11)
zl = rng.standard_normal((40, 3))
wide = [zl @ rng.standard_normal((3, d)) + 0.5 * rng.standard_normal((40, d)) for d in (60, 55)]
wide_f = [zl @ rng.standard_normal((3, d)) + 0.5 * rng.standard_normal((40, d)) for d in (60, 55)]
linear_case("wide_two_view_f64", wide, wide_f, {
    "rcca_c0.3": lambda: rCCA(latent_dimensions=3, c=0.3),
    "mcca_c0.3": lambda: MCCA(latent_dimensions=3, c=0.3),
    "mcca_c0.3_nopca": lambda: MCCA(latent_dimensions=3, c=0.3, pca=False),
    "gcca_c0.3": lambda: GCCA(latent_dimensions=3, c=0.3),
})

# ---- function seams: svd_whiten / gevp --------------------------------------
rng = np.random.default_rng(3)
Xw = rng.standard_normal((120, 14))
Xw -= Xw.mean(axis=0)
store = {"X": Xw}
for c in (0.0, 0.25, 1.0):
    xw, W = svd_whiten(Xw, c)
    store[f"c{c}/W"] = W
    store[f"c{c}/X_white_head"] = xw[:5]
A = rng.standard_normal((30, 30))
A = A + A.T
Bm = rng.standard_normal((30, 30))
Bm = Bm @ Bm.T + 30 * np.eye(30)
store["A"], store["B"] = A, Bm
w, V = gevp(A, None, 5)
store["gevp_std/w"], store["gevp_std/V"] = w, V
w, V = gevp(A, Bm, 5)
store["gevp_gen/w"], store["gevp_gen/V"] = w, V
np.savez_compressed(os.path.join(OUT, "linalg_seams.npz"), **store)
print("linalg_seams", len(store), "arrays")

# ---- DCCA losses: value + input gradients -----------------------------------
store = {}
for (n, d) in [(64, 8), (256, 32), (1024, 48)]:
    for dt, dname in [(torch.float32, "f32"), (torch.float64, "f64")]:
        for eps in (1e-6, 1e-4):
            if n == 1024 and eps != 1e-6:
                continue                      # keep the fixture file small
            torch.manual_seed(0)
            z1 = torch.randn(n, d, dtype=dt)
            z2 = (0.5 * z1[:, : d] + torch.randn(n, d, dtype=dt))
            z1.requires_grad_(True)
            z2.requires_grad_(True)
            loss = CCALoss(eps=eps)([z1, z2])
            loss.backward()
            tag = f"cca/n{n}_d{d}_{dname}_eps{eps:g}"
            store[tag + "/z1"] = z1.detach().numpy()
            store[tag + "/z2"] = z2.detach().numpy()
            store[tag + "/loss"] = loss.detach().numpy()
            store[tag + "/g1"] = z1.grad.numpy()
            store[tag + "/g2"] = z2.grad.numpy()
# unequal widths
torch.manual_seed(1)
z1 = torch.randn(300, 12, dtype=torch.float64, requires_grad=True)
z2 = torch.randn(300, 7, dtype=torch.float64, requires_grad=True)
loss = CCALoss(eps=1e-5)([z1, z2])
loss.backward()
for k, v in dict(z1=z1, z2=z2, loss=loss, g1=z1.grad, g2=z2.grad).items():
    store["cca/unequal/" + k] = v.detach().numpy()
# three-view MCCALoss
torch.manual_seed(2)
base = torch.randn(200, 6, dtype=torch.float64)
zs3 = [(base + 0.7 * torch.randn(200, 6, dtype=torch.float64)).requires_grad_(True) for _ in range(3)]
loss = MCCALoss(eps=1e-5)(zs3)
loss.backward()
for i, z in enumerate(zs3):
    store[f"mcca/z{i}"] = z.detach().numpy()
    store[f"mcca/g{i}"] = z.grad.numpy()
store["mcca/loss"] = loss.detach().numpy()
# _inv_sqrtm
torch.manual_seed(3)
a = torch.randn(40, 16, dtype=torch.float64)
spd = a.T @ a / 39
store["inv_sqrtm/A"] = spd.numpy()
store["inv_sqrtm/out_eps1e-5"] = _inv_sqrtm(spd, 1e-5).numpy()
store["inv_sqrtm/out_eps0.5"] = _inv_sqrtm(spd, 0.5).numpy()      # clamp active
np.savez_compressed(os.path.join(OUT, "losses.npz"), **store)
print("losses", len(store), "arrays")

# ------------------------------------------------------------------------------------------------
# model selection: the reference's GridSearchCV, one refit per (setting, fold)
# ------------------------------------------------------------------------------------------------
from cca_zoo.model_selection import GridSearchCV  # noqa: E402

store = {}
rng = np.random.default_rng(11)
z = rng.standard_normal((240, 3)) * np.array([2.0, 1.3, 0.8])
gv = [z @ rng.standard_normal((3, p)) + 0.9 * rng.standard_normal((240, p)) + off
      for p, off in ((12, 0.5), (9, -1.0), (7, 0.0))]
for i, v in enumerate(gv):
    store[f"view{i}"] = v


def grid_case(tag, est, grid, views, cv):
    gs = GridSearchCV(est, param_grid=grid, cv=cv).fit(views)
    res = gs.cv_results_
    store[f"{tag}/mean_test_score"] = np.asarray(res["mean_test_score"], dtype=np.float64)
    store[f"{tag}/std_test_score"] = np.asarray(res["std_test_score"], dtype=np.float64)
    store[f"{tag}/rank_test_score"] = np.asarray(res["rank_test_score"], dtype=np.int64)
    for f in range(cv):
        store[f"{tag}/split{f}_test_score"] = np.asarray(res[f"split{f}_test_score"], dtype=np.float64)
    store[f"{tag}/best_score"] = np.float64(gs.best_score_)
    store[f"{tag}/best_index"] = np.int64(gs._inner_cv.best_index_)
    for i, w in enumerate(gs.best_estimator_.weights_):
        store[f"{tag}/best_w{i}"] = np.asarray(w)
    store[f"{tag}/score_all"] = np.float64(gs.score(views))
    store[f"{tag}/params"] = np.array([repr(sorted(p.items())) for p in res["params"]])


grid_case("rcca", rCCA(), {"c": [0.0, 0.01, 0.1, 0.5, 0.9], "latent_dimensions": [1, 2]}, gv[:2], 4)
grid_case("mcca", MCCA(), {"c": [0.0, 0.1, 0.7], "latent_dimensions": [2]}, gv, 3)
grid_case("gcca", GCCA(), {"c": [0.05, 0.3], "latent_dimensions": [1, 2]}, gv, 3)
np.savez_compressed(os.path.join(OUT, "grid_search.npz"), **store)
print("grid_search", len(store), "arrays")

# ------------------------------------------------------------------------------------------------
# PartialCCA / GRCCA: MCCA hooks on deconfounded / group-augmented views
# ------------------------------------------------------------------------------------------------
from cca_zoo.linear import GRCCA, PartialCCA  # noqa: E402

store = {}
rng = np.random.default_rng(21)
n = 180
conf = rng.standard_normal((n, 3)) + np.array([0.5, -0.2, 1.0])          # confounds with non-zero means
lat = rng.standard_normal((n, 2)) * np.array([1.8, 1.1])
pv = [lat @ rng.standard_normal((2, p)) + conf @ rng.standard_normal((3, p)) + 0.8 * rng.standard_normal((n, p)) + off
      for p, off in ((10, 1.0), (7, -0.5), (6, 0.0))]
store["partials"] = conf
for i, v in enumerate(pv):
    store[f"view{i}"] = v
for tag, kw, vs in (("pcca_2v", dict(latent_dimensions=2), pv[:2]),
                    ("pcca_3v_ridge", dict(latent_dimensions=2, c=[0.1, 0.3, 0.0]), pv),
                    ("pcca_nocenter", dict(latent_dimensions=1, center=False, c=0.2), pv[:2])):
    m = PartialCCA(**kw).fit(vs, partials=conf)
    for i, w in enumerate(m.weights_):
        store[f"{tag}/w{i}"] = np.asarray(w)
    for i, b in enumerate(m.confound_betas_):
        store[f"{tag}/beta{i}"] = np.asarray(b)
    for i, mu in enumerate(m.means_):
        store[f"{tag}/mean{i}"] = np.asarray(mu)
    for i, t in enumerate(m.transform(vs, partials=conf)):
        store[f"{tag}/transform_partials{i}"] = t[:6]
    for i, t in enumerate(m.transform(vs)):
        store[f"{tag}/transform_plain{i}"] = t[:6]
    store[f"{tag}/score"] = m.score(vs)
groups = [np.array([0, 0, 0, 1, 1, 2, 2, 2, 2, 3]), np.array([5, 5, 7, 7, 7, 9, 9]), np.array([1, 1, 1, 2, 2, 2])]
for i, gidx in enumerate(groups):
    store[f"groups{i}"] = gidx
for tag, kw, vs, gs in (("grcca_2v", dict(latent_dimensions=2, c=[0.5, 0.8], mu=[0.3, 0.0]), pv[:2], groups[:2]),
                        ("grcca_3v_mixed", dict(latent_dimensions=2, c=[0.4, 0.0, 0.9], mu=[1.5, 0.2, 0.0]), pv, groups)):
    m = GRCCA(**kw).fit(vs, feature_groups=gs)
    for i, w in enumerate(m.weights_):
        store[f"{tag}/w{i}"] = np.asarray(w)
    store[f"{tag}/score"] = m.score(vs)
    for i, t in enumerate(m.transform(vs)):
        store[f"{tag}/transform{i}"] = t[:6]
np.savez_compressed(os.path.join(OUT, "partial_group.npz"), **store)
print("partial_group", len(store), "arrays")

# ------------------------------------------------------------------------------------------------
# GCCALoss and _BatchWhiten
# ------------------------------------------------------------------------------------------------
import importlib.machinery  # noqa: E402

from cca_zoo.deep.objectives import GCCALoss  # noqa: E402

if "lightning" not in sys.modules:      # shim: _dcca_noi.py imports the Lightning base class only to subclass it
    _l = types.ModuleType("lightning")
    _l.__spec__ = importlib.machinery.ModuleSpec("lightning", None)
    _lp = types.ModuleType("lightning.pytorch")
    _lp.__spec__ = importlib.machinery.ModuleSpec("lightning.pytorch", None)
    _lp.LightningModule = torch.nn.Module
    _l.pytorch = _lp
    sys.modules["lightning"] = _l
    sys.modules["lightning.pytorch"] = _lp
from cca_zoo.deep._dcca_noi import _BatchWhiten  # noqa: E402

store = {}
torch.manual_seed(7)
base = torch.randn(300, 5, dtype=torch.float64)
for tag, dims, eps in (("gcca3", (5, 5, 5), 1e-5), ("gcca2_eps", (5, 5), 1e-2), ("gcca4", (5, 5, 5, 5), 1e-4)):
    zs = [(base @ torch.randn(5, d, dtype=torch.float64) + 0.8 * torch.randn(300, d, dtype=torch.float64) + 0.3 * i)
          .requires_grad_(True) for i, d in enumerate(dims)]
    loss = GCCALoss(eps=eps)(zs)
    loss.backward()
    for i, z in enumerate(zs):
        store[f"{tag}/z{i}"] = z.detach().numpy()
        store[f"{tag}/g{i}"] = z.grad.numpy()
    store[f"{tag}/loss"] = loss.detach().numpy()
    store[f"{tag}/eps"] = np.float64(eps)
# _BatchWhiten: three training steps (running covariance EMA), gradient of a fixed functional, eval passthrough
bw = _BatchWhiten(6, momentum=0.2, eps=1e-4).double()
bw.train()
torch.manual_seed(8)
mix = torch.randn(6, 6, dtype=torch.float64)
for step in range(3):
    x = (torch.randn(120, 6, dtype=torch.float64) @ mix + 0.1 * step).requires_grad_(True)
    y = bw(x)
    coef = torch.linspace(0.5, 1.5, 6, dtype=torch.float64)
    ((y * y) @ coef).sum().backward()
    store[f"bw/x{step}"] = x.detach().numpy()
    store[f"bw/y{step}"] = y.detach().numpy()
    store[f"bw/gx{step}"] = x.grad.numpy()
    store[f"bw/running{step}"] = bw.running_covar.detach().numpy().copy()
bw.eval()
store["bw/eval_identity"] = np.float64(float((bw(x.detach()) - x.detach()).abs().max()))
store["bw/num_batches"] = np.int64(int(bw.num_batches_tracked))
np.savez_compressed(os.path.join(OUT, "deep_next.npz"), **store)
print("deep_next", len(store), "arrays")

# ---- fp32 views whose means dwarf their spread (mean = 100 sigma): the reference centres BEFORE any product
# (cca_zoo/_base.py:97-99), so its fp32 thin SVD is unaffected; a second-moment formulation must subtract a pilot
# mean to match it (VERDICT r1 item 2 / ADVICE r1) ------------------------------------------------------------------
rng_off = np.random.default_rng(7)
n_off, k_off = 4000, 4
z_off = rng_off.standard_normal((n_off, k_off)) * np.linspace(2.0, 0.5, k_off)
off100_train, off100_fresh = [], []
for d_off in (40, 30):
    load = rng_off.standard_normal((k_off, d_off))
    x = z_off @ load + rng_off.standard_normal((n_off, d_off))
    shift = 100.0 * x.std(axis=0) * rng_off.choice([-1.0, 1.0], size=d_off)
    off100_train.append((x + shift).astype(np.float32))
    y = (rng_off.standard_normal((500, k_off)) * np.linspace(2.0, 0.5, k_off)) @ load + rng_off.standard_normal((500, d_off))
    off100_fresh.append((y + shift).astype(np.float32))
linear_case("offset_two_view_f32", off100_train, off100_fresh, {
    "rcca_0.1": lambda: rCCA(latent_dimensions=4, c=0.1),
    "cca": lambda: CCA(latent_dimensions=4),
})
