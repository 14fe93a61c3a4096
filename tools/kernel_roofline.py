#!/usr/bin/env python3
"""Per-launch roofline of ONE training step: joins the step timeline (tools/trace_step.py output: one row per kernel node),
the per-dispatch PMC traffic of the same build (tools/pmc_traffic.py --step) and an algorithmic model of every launch
(layer, bytes each operand once, FLOPs) into a CSV: us, TB/s, PFLOP/s, the binding roof and the fraction of it, wasted traffic.

    python tools/kernel_roofline.py profiles/r03_step_timeline.txt [profiles/r03_pmc_step_traffic.json] \
        --out profiles/r03_kernel_roofline.csv [--size 96 --batch 2]

Model: UNet3D(1 -> 4, features 16/32/64/128), bf16 activations, fp32 input / logits, int64 labels; BatchNorm
statistics ride in the conv epilogue (their launches carry no algorithmic bytes of their own); apply = R y + W z (+ pooled);
BN backward = reduce (R dz, R y) + apply (R dz, R y, W dy); a conv backward reads dy for each of its two products, x once,
writes dx; weights fp32 once.  The walker assigns launches to layers from the kernel NAMES in launch order (the plan's order:
encoder 0..3, bottleneck, decoder 0..3, head; backward in reverse), so it follows route changes as long as the kernel families
keep their names.  Roofs: HBM 8 TB/s, dense bf16 MFMA 2.5 PFLOP/s; a launch is priced against the roof that binds it."""
import csv
import json
import re
import sys

HBM, MFMA = 8.0e12, 2.5e15


def parse_timeline(path):
    rows = []
    for line in open(path):
        m = re.match(r"\s*(\d+)\s+([\d.]+)\s+([\d.]+)\s+(-?[\d.]+)\s+(.*?)\s+grid (\d+) wg (\d+)", line)
        if m:
            rows.append({"idx": int(m.group(1)), "start": float(m.group(2)), "us": float(m.group(3)), "name": m.group(5).strip(),
                         "grid": int(m.group(6))})
    return rows


class Net:
    def __init__(self, size, batch, feats=(16, 32, 64, 128)):
        self.L = len(feats)
        self.C = list(feats) + [2 * feats[-1]]
        self.M = [batch * (size >> l) ** 3 for l in range(self.L + 1)]
        blocks = []                                  # (name, level, cin, cout)
        for l in range(self.L):
            blocks.append((f"encoder.{l}", l, 1 if l == 0 else self.C[l - 1], self.C[l]))
        blocks.append(("bottleneck", self.L, self.C[self.L - 1], self.C[self.L]))
        for i in range(self.L):
            l = self.L - 1 - i
            blocks.append((f"decoder.{i}", l, 2 * self.C[l], self.C[l]))
        self.blocks = blocks
        self.halves = []                             # forward order: (layer name, level, cin, cout)
        for name, l, ci, co in blocks:
            self.halves.append((name + ".conv0", l, ci, co))
            self.halves.append((name + ".conv1", l, co, co))
        self.params = sum(27 * ci * co + 3 * co for _, _, ci, co in self.halves) + sum(
            8 * 2 * self.C[l] * self.C[l] + self.C[l] for l in range(self.L)) + 4 * self.C[0] + 4


def conv_bytes(M, ci, co, first):
    return M * ci * (4 if first else 2) + M * co * 2 + 27 * ci * co * 4


def model(rows, net):
    """Yield (layer, what, algo_bytes, flops) per row."""
    out = []
    fi = -1            # index into net.halves, forward
    up_f = -1
    phase = "fwd"
    bi = len(net.halves)     # backward: next half index (counts down)
    up_b = net.L
    cur_b = None
    pool_b = net.L
    for r in rows:
        n = r["name"]
        layer, what, by, fl = "?", "?", 0.0, 0.0
        if "pack_all" in n:
            layer, what = "all weights", "pack fp32 -> 2 bf16 MFMA images"
            conv_w = sum(27 * ci * co for _, _, ci, co in net.halves[1:])
            by = conv_w * 4 + 2 * conv_w * 2 * 28 / 27
        elif phase == "fwd" and ("conv3_c1_fwd" in n or "conv3_mfma_persist" in n or re.match(r"conv3_mfma8?_kernel", n)):
            fi += 1
            name, l, ci, co = net.halves[fi]
            layer, what = name, "conv3 fwd" + (" (split-K partials)" if "false, true>" in n else "")
            by, fl = conv_bytes(net.M[l], ci, co, fi == 0), 2.0 * 27 * ci * co * net.M[l]
        elif "bn_stats_finalize" in n or "bn_stats_splitk" in n or "bn_stats_kernel" in n:
            name, l, ci, co = net.halves[fi]
            layer, what = name, "BN statistics" + (" + split-K finish" if "splitk" in n else " finalize")
        elif "bn_apply_pool" in n:
            name, l, ci, co = net.halves[fi]
            layer, what, by = name, "BN apply + ReLU + MaxPool", net.M[l] * co * 2 * (2 + 1 / 8)
        elif "bn_apply" in n:
            name, l, ci, co = net.halves[fi]
            layer, what, by = name, "BN apply + ReLU", net.M[l] * co * 2 * 2
        elif "upconv_mfma_fwd" in n or n.startswith("upconv2_fwd"):
            up_f += 1
            l = net.L - 1 - up_f
            layer, what = f"upconvs.{up_f}", "ConvTranspose fwd"
            by = net.M[l + 1] * 2 * net.C[l] * 2 + net.M[l] * net.C[l] * 2 + 8 * 2 * net.C[l] * net.C[l] * 4
            fl = 2.0 * 8 * 2 * net.C[l] * net.C[l] * net.M[l + 1]
        elif "head_loss_fwd" in n:
            layer, what = "final_conv + loss", "1x1x1 conv + Dice/CE + metrics fwd (logits never written)"
            by, fl = net.M[0] * (net.C[0] * 2 + 8), 2.0 * net.C[0] * 4 * net.M[0]
        elif "head_loss_bwd" in n:
            phase = "bwd"
            layer, what = "final_conv + loss", "loss bwd + 1x1x1 conv bwd (dlogits never written)"
            by, fl = net.M[0] * (net.C[0] * 2 + 8 + net.C[0] * 2), 2.0 * 3 * net.C[0] * 4 * net.M[0]
        elif "conv1_fwd" in n:
            layer, what, by, fl = "final_conv", "1x1x1 conv fwd", net.M[0] * (net.C[0] * 2 + 4 * 4), 2.0 * net.C[0] * 4 * net.M[0]
        elif "seg_loss_fwd" in n:
            layer, what, by = "loss", "Dice/CE + metrics fwd", net.M[0] * (4 * 4 + 8)
        elif "seg_loss_metrics_finalize" in n:
            layer, what = "loss", "finalize"
        elif "seg_loss_bwd" in n:
            phase = "bwd"
            layer, what, by = "loss", "loss bwd (dlogits)", net.M[0] * (4 * 4 + 8 + 4 * 4)
        elif "conv1_bwd" in n:
            layer, what = "final_conv", "1x1x1 conv bwd"
            by, fl = net.M[0] * (net.C[0] * 2 + 4 * 4 + net.C[0] * 2), 2.0 * 2 * net.C[0] * 4 * net.M[0]
        elif "bn_bwd_reduce" in n or "bn_bwd_onepass" in n:
            bi -= 1
            cur_b = net.halves[bi]
            name, l, ci, co = cur_b
            one = "onepass" in n
            layer, what, by = name, "BN bwd " + ("one pass" if one else "reduce") + (" (+ slab sum)" if "slab" in n else ""), \
                net.M[l] * co * 2 * (3 if one else 2)
        elif "bn_bwd_finalize" in n:
            layer, what = cur_b[0], "BN bwd finalize"
        elif "bn_bwd_apply" in n:
            name, l, ci, co = cur_b
            layer, what, by = name, "BN bwd apply", net.M[l] * co * 2 * 3
        elif "conv3_bwd_fused" in n or "conv3_wgrad" in n:
            name, l, ci, co = cur_b
            first = bi == 0
            layer, what = name, "conv3 bwd (dgrad + wgrad)" if not first else "conv3 wgrad (first layer)"
            by = net.M[l] * co * 2 * (1 if first else 2) + net.M[l] * ci * (4 if first else 2) + (0 if first else net.M[l] * ci * 2) + 27 * ci * co * 4
            fl = 2.0 * 27 * ci * co * net.M[l] * (1 if first else 2)
        elif "slab_job" in n or "bwd_tail" in n or "slab_reduce" in n or "splitk_finish" in n:
            layer, what = (cur_b[0] if cur_b else "?"), "slab sum / split-K finish"
        elif "upconv_mfma_bwd" in n or n.startswith("upconv2_bwd"):
            up_b -= 1
            l = net.L - 1 - up_b
            layer, what = f"upconvs.{up_b}", "ConvTranspose bwd (dgrad + wgrad)"
            by = 2 * net.M[l] * net.C[l] * 2 + 2 * net.M[l + 1] * 2 * net.C[l] * 2 + 8 * 2 * net.C[l] * net.C[l] * 4
            fl = 2.0 * 2 * 8 * 2 * net.C[l] * net.C[l] * net.M[l + 1]
        elif "maxpool2_bwd" in n:
            pool_b -= 1
            l = pool_b
            layer, what, by = f"pool.{l}", "MaxPool bwd + skip add", net.M[l] * net.C[l] * 2 * (3 + 1 / 8)
        elif "adamw" in n:
            layer, what, by = "optimizer", "AdamW", 7 * 4 * net.params
        elif "step_inc" in n:
            layer, what = "optimizer", "step counter"
        elif "copyBuffer" in n or "fill" in n.lower() or "memset" in n.lower():
            layer, what = "runtime", "copy / fill"
        out.append((layer, what, by, fl))
    return out


def main():
    args = [a for a in sys.argv[1:] if not a.startswith("--")]
    opt = {sys.argv[i][2:]: sys.argv[i + 1] for i in range(1, len(sys.argv) - 1) if sys.argv[i].startswith("--")}
    rows = parse_timeline(args[0])
    pmc = json.load(open(args[1]))["per_dispatch"] if len(args) > 1 else None
    net = Net(int(opt.get("size", 96)), int(opt.get("batch", 2)))
    mod = model(rows, net)
    if pmc is not None and len(pmc) != len(rows):
        print(f"warning: {len(pmc)} PMC dispatches vs {len(rows)} timeline rows: traffic column left empty", file=sys.stderr)
        pmc = None
    out = opt.get("out", "kernel_roofline.csv")
    tot_us = sum(r["us"] for r in rows)
    with open(out, "w", newline="") as f:
        w = csv.writer(f)
        w.writerow(["idx", "kernel", "layer", "what", "us", "share_of_step", "algo_MB", "GFLOP", "TB_per_s", "PFLOP_per_s", "bound",
                    "frac_of_bound", "pmc_MB", "pmc_over_algo"])
        agg = {}
        for i, (r, (layer, what, by, fl)) in enumerate(zip(rows, mod)):
            t = r["us"] * 1e-6
            tb, pf = by / t / 1e12, fl / t / 1e15
            bound = "-"
            frac = 0.0
            if by > 0 or fl > 0:
                bound = "mfma" if fl / MFMA > by / HBM else "hbm"
                frac = (fl / MFMA if bound == "mfma" else by / HBM) / t
            pm = ""
            ratio = ""
            if pmc is not None:
                b = pmc[i]["fetch_bytes_corrected"] + pmc[i]["write_bytes"]
                pm = f"{b / 1e6:.1f}"
                ratio = f"{b / by:.2f}" if by > 0 else ""
            w.writerow([r["idx"], r["name"][:60], layer, what, f"{r['us']:.1f}", f"{r['us'] / tot_us:.4f}", f"{by / 1e6:.1f}", f"{fl / 1e9:.2f}",
                        f"{tb:.2f}", f"{pf:.3f}", bound, f"{frac:.3f}", pm, ratio])
            lvl = "other"
            m = re.match(r"(encoder|decoder)\.(\d)", layer)
            if m:
                lvl = f"level {int(m.group(2)) if m.group(1) == 'encoder' else net.L - 1 - int(m.group(2))}"
            elif layer.startswith("bottleneck"):
                lvl = f"level {net.L}"
            elif layer.startswith("upconvs"):
                lvl = f"level {net.L - 1 - int(layer.split('.')[1])}"
            elif layer.startswith("pool"):
                lvl = f"level {layer.split('.')[1]}"
            elif layer in ("final_conv", "loss"):
                lvl = "level 0"
            a = agg.setdefault(lvl, [0.0, 0.0, 0.0, 0])
            a[0] += r["us"]; a[1] += by; a[2] += fl; a[3] += 1
    unknown = sum(1 for m_ in mod if m_[0] == "?")
    print(f"{len(rows)} launches, {tot_us:.1f} us, {unknown} unassigned -> {out}")
    for lvl in sorted(agg):
        us, by, fl, n = agg[lvl]
        print(f"  {lvl:8s} {n:3d} launches {us:8.1f} us ({us / tot_us * 100:4.1f} %)  {by / 1e6:8.1f} MB (HBM floor {by / HBM * 1e6:6.1f} us)  "
              f"{fl / 1e9:7.1f} GFLOP (MFMA floor {fl / MFMA * 1e6:5.1f} us)")


if __name__ == "__main__":
    main()
