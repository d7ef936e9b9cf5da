"""The experiment behind the blend kernels' tile order: their workgroups handed out heaviest tile first (LPT) instead of
in the spatial order - (a) globally (exact argsort on the host), (b) inside each XCD's share of the spatial order (the
tile-to-XCD assignment, and with it which L2 serves a tile, unchanged), (c) as the library computes it (cugs_tile_order:
513 length classes).  Uses the public `tile_order` argument of rasterize_forward / rasterize_backward.
    python tools/lpt_order.py [uniform | FRAC:AREA ...]"""
import ctypes as C
import os
import sys

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import __graft_entry__ as ge

pkg = ge.load_package()
R = pkg.rasterizer
dev = torch.device("cuda", 0)
wl = pkg.scene.CONFIGS["config3"]
ntx, nty = (wl.width + 15) // 16, (wl.height + 15) // 16
tiles = ntx * nty
G = 2                                             # CUGS_ROW_GROUP


def shipped_tile(bid):
    """cugs_blend_tile (csrc/cugs_common.h) restated."""
    nwg = tiles
    xcd, q, r = bid & 7, nwg >> 3, nwg & 7
    base = xcd * (q + 1) if xcd < r else r * (q + 1) + (xcd - r) * q
    lin = base + (bid >> 3)
    row, col = divmod(lin, ntx)
    nunits = (nty + G - 1) // G
    c, urow = 0, row
    for _ in range(7):
        units = (nunits + 7 - c) >> 3
        last_in = 1 if (units > 0 and ((nunits - 1) & 7) == c) else 0
        rows_c = units * G - last_in * (nunits * G - nty)
        if urow >= rows_c:
            urow -= rows_c
            c += 1
    unit = c + 8 * (urow // G)
    return (unit * G + urow % G) * ntx + col


SHIPPED = np.array([shipped_tile(b) for b in range(tiles)], dtype=np.int64)
assert sorted(SHIPPED.tolist()) == list(range(tiles))


def time_kernel(fn, reps=15):
    ts = []
    for _ in range(reps):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record(); fn(); e1.record(); torch.cuda.synchronize()
        ts.append(e0.elapsed_time(e1) * 1000.0)
    ts.sort()
    return ts[len(ts) // 2]


for spec in (sys.argv[1:] or ["uniform", "0.8:0.1", "0.5:0.02"]):
    cluster = None if spec == "uniform" else tuple(float(x) for x in spec.split(":"))
    arrays = pkg.scene.make_gaussians(wl.n, wl.width, wl.height, sh_degree=wl.sh_degree, cluster=cluster)
    cam = pkg.scene.make_camera(wl.width, wl.height)
    model = pkg.scene.to_model(arrays, dev)
    st = pkg.RenderSettings(active_sh_degree=wl.sh_degree)
    g = torch.from_numpy(pkg.scene.make_dl_dcolor(wl.width, wl.height)).to(dev)
    out = pkg.render(model, cam, st)
    lens = (out.tile_ranges[:, 1] - out.tile_ranges[:, 0]).cpu().numpy().astype(np.int64)
    orders = {"spatial": None}
    orders["lpt-global"] = np.argsort(-lens, kind="stable")
    per = np.empty(tiles, dtype=np.int64)
    for x in range(8):
        mine = SHIPPED[x::8]
        per[x::8] = mine[np.argsort(-lens[mine], kind="stable")]
    orders["lpt-per-xcd"] = per
    orders["library"] = R.tile_order_of(out.tile_ranges, wl.width, wl.height)[:, 0].cpu().numpy().astype(np.int64)
    tr_host = out.tile_ranges.cpu().numpy()

    def records(perm):                            # {tile, first pair, one past the last pair, 0}
        rec = np.zeros((tiles, 4), dtype=np.int32)
        rec[:, 0] = perm
        rec[:, 1:3] = tr_host[perm]
        return torch.from_numpy(rec).to(dev)
    cur = {"order": None}
    fwd = lambda: R.rasterize_forward(out.means_2d, out.cov_2d_inv, out.rgb, out.opacities_act, out.tile_ranges,
                                      out.gaussian_indices, wl.width, wl.height, st.background, packed=out.packed,
                                      tile_order=cur["order"])
    bwd = lambda: R.rasterize_backward(g, out.means_2d, out.cov_2d_inv, out.rgb, out.opacities_act, out.tile_ranges,
                                       out.gaussian_indices, out.final_T, out.n_contrib, wl.width, wl.height, st.background,
                                       wl.n, packed=out.packed, unpack=False, tile_order=cur["order"])
    ref_color = fwd().color.clone()
    print(f"{spec}: pairs {out.total_pairs}, tile lists mean {lens.mean():.0f} max {lens.max()}", flush=True)
    for rnd in range(2):
        for name, perm in orders.items():
            cur["order"] = None if perm is None else records(perm)
            torch.cuda.synchronize()
            assert torch.equal(fwd().color, ref_color)
            tf, tb = time_kernel(fwd), time_kernel(bwd)
            print(f"   {name:12s} forward {tf:7.1f} us   backward (with the accumulator fill) {tb:7.1f} us", flush=True)
