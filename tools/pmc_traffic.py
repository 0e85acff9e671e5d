"""HBM traffic per launch of the dominant convolution from rocprofv3 PMC passes -> profiles/r06/pmc_traffic.json
(read by bench.py for roofline.traffic, which refuses it once the kernel sources change).  Round 3: the dominant launch
is the Winograd kernel (wino_conv8s_kernel); the direct kernel and the weight gradient are recorded next to it.

    # on the GPU box, two counter-only passes (MI355X_MICROARCH.md: FETCH_SIZE and WRITE_SIZE do not fit one pass):
    rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d gpurun_out/pmc_fetch -- python3 tools/bench_wino.py --rounds 1 --iters 3 --shapes 256,256,32
    rocprofv3 --kernel-trace --pmc WRITE_SIZE --output-format csv -d gpurun_out/pmc_write -- python3 tools/bench_wino.py --rounds 1 --iters 3 --shapes 256,256,32
    python tools/pmc_traffic.py gpurun_out/pmc_fetch gpurun_out/pmc_write [out.json]

Corrections as the guide prescribes for gfx950: counter unit KB -> bytes; FETCH_SIZE x 2 (128-byte requests tallied
at 64 bytes); WRITE_SIZE as is."""
import csv
import glob
import hashlib
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
SOURCES = ["psld_amd/csrc/conv_wino.hip", "psld_amd/csrc/conv_split.hip", "psld_amd/csrc/limb.h", "psld_amd/csrc/tile_shared.h",
           "psld_amd/csrc/common.h"]
KERNELS = {"wino": "wino_conv8s_kernel", "dconv": "dconv_lp_kernel<1", "dconv_f32": "dconv_kernel<7", "dwgrad": "dwgrad_kernel<4"}


def per_launch(directory, counter):
    out = {}
    files = glob.glob(os.path.join(directory, "**", "*counter_collection.csv"), recursive=True)
    assert files, f"no *counter_collection.csv under {directory}"
    for f in files:
        with open(f, newline="") as fh:
            for row in csv.DictReader(fh):
                if row.get("Counter_Name") != counter:
                    continue
                name = row["Kernel_Name"]
                for tag, pat in KERNELS.items():
                    if pat in name:
                        out.setdefault(tag, []).append(float(row["Counter_Value"]))
    return {k: (sum(v) / len(v), len(v)) for k, v in out.items()}


def sources_sha():
    h = hashlib.sha256()
    for f in SOURCES:
        with open(os.path.join(ROOT, f), "rb") as fh:
            h.update(fh.read())
    return h.hexdigest()


def main():
    fetch = per_launch(sys.argv[1], "FETCH_SIZE")
    write = per_launch(sys.argv[2], "WRITE_SIZE")
    f_kb, nf = fetch["wino"]
    w_kb, nw = write["wino"]
    fetch_b, write_b = f_kb * 1024 * 2, w_kb * 1024
    # input + output + residual fp32 (the benchmark's epilogue reads a residual), Winograd limb fragments 16 x 6 B / weight
    algo = 3 * 128 * 32 * 32 * 256 * 4 + 256 * 256 * 16 * 6
    rec = {
        "source": "rocprofv3 --kernel-trace --pmc FETCH_SIZE and --pmc WRITE_SIZE (separate counter-only passes) on "
                  "tools/bench_wino.py --rounds 1 --iters 3 --shapes 256,256,32; FETCH_SIZE doubled per MI355X_MICROARCH.md HBM "
                  "section (gfx950 tallies 128-B requests at 64 B); KB -> bytes",
        "kernel": "wino_conv8s_kernel (3x3 conv forward, Winograd F(2x2,3x3), bf16x6 limb MFMA, fp32 input, bias + residual epilogue)",
        "shape": "conv3x3 256->256, 32x32, B=128 (M=131072, N=256, K=2304): the most frequent heavy launch of the step",
        "launches_averaged": [nf, nw],
        "fetch_size_kb_raw": f_kb, "fetch_bytes_corrected": fetch_b, "write_bytes": write_b,
        "traffic_bytes": fetch_b + write_b, "algorithmic_bytes": algo,
        "ratio": (fetch_b + write_b) / algo,
        "sources": SOURCES, "sources_sha256": sources_sha(),
        "note": f"HBM bytes/launch of the conv3x3 256->256 @32x32 B=128 Winograd launch (PMC FETCH_SIZE x2 + WRITE_SIZE, "
                f"this build): {(fetch_b + write_b) / 1e6:.1f} MB vs {algo / 1e6:.1f} MB algorithmic "
                f"(input 134.2 + residual 134.2 + output 134.2 + Winograd limb fragments 6.3).  STRUCTURAL for this kernel's "
                f"128-channel accumulator tile, and closed as such (VERDICT r04 #6b): the input is fetched once per 128-channel "
                f"output tile (2 of them) x 6/4 halo rows = 3 x 134.2 = 402.6 MB, + residual 134.2 + the transformed weights "
                f"streamed once per XCD (~25 MB) = 562 MB of the {fetch_b / 1e6:.0f} MB fetched (the rest: 34-pixel halo rows "
                f"over-fetching partial lines).  The two channel tiles of a pixel tile run on DIFFERENT XCDs (channel-tile-major "
                f"order: an XCD streams ONE 3.1 MB slice of U, which stays in its 4 MB L2); putting them on the same XCD "
                f"(pixel-tile-major) was measured in round 3: 690 MB fetched (both slices of U no longer fit the L2) and 1-2 % "
                f"slower.  At {(fetch_b + write_b) / 1e6 / 0.44:.0f} GB/s over the launch's ~440 us this is 0.2 of the HBM rate: not the bound.",
    }
    for tag, label in (("dconv", "dconv_lp_kernel (direct, limb-plane input, same shape and epilogue)"),
                       ("dconv_f32", "dconv_kernel<7,9> (direct, fp32 input, same shape and epilogue)"),
                       ("dwgrad", "dwgrad_kernel<4> 256->256 @32x32 B=128")):
        if tag in fetch and tag in write:
            rec[label] = {"fetch_bytes_corrected": fetch[tag][0] * 2048, "write_bytes": write[tag][0] * 1024}
    path = sys.argv[3] if len(sys.argv) > 3 else os.path.join(ROOT, "profiles", "r06", "pmc_traffic.json")
    os.makedirs(os.path.dirname(path), exist_ok=True)
    with open(path, "w") as fh:
        json.dump(rec, fh, indent=1)
    print(json.dumps(rec, indent=1))


if __name__ == "__main__":
    main()
