#!/usr/bin/env python3
"""HBM traffic of ONE forward of the BERT feature extractor (bench.py's `bert_zh_features` leg) from rocprofv3 PMC counters (run ON
THE GPU BOX).  Same method as tools/collect_traffic.py (MI355X_MICROARCH.md, HBM / rocprofv3 PMC sections): FETCH_SIZE and WRITE_SIZE
in SEPARATE `rocprofv3 --kernel-trace --pmc <counter>` passes, KB -> bytes, FETCH_SIZE doubled (gfx950 wide-read correction),
WRITE_SIZE as reported.  All kernels of a forward are summed; forwards are counted by the embedding kernel's dispatches.

    python tools/collect_traffic_bert.py profiles/r02_e_traffic_bert.json
"""
import collections, csv, glob, hashlib, json, os, subprocess, sys, tempfile

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
SCRIPT = """
import sys, torch
sys.path.insert(0, %r)
from bert_vits2_amd.bert_encoder import BertEncoder
from bert_vits2_amd import bert_synth as BS
cfg = BS.LARGE
enc = BertEncoder(**cfg).load_state_dict(BS.bert_state_dict(cfg, 0, layers=22), device="cuda:0")
ids, _ = BS.synthetic_inputs(cfg, [53], 0)          # bench.py's bert_zh_features workload: B = 1, S = 53
ids = ids.cuda()
for _ in range(10):
    enc(ids)
torch.cuda.synchronize()
""" % ROOT
SOURCES = ("bert.hip", "conv_mfma.hip", "attention.hip")


def one_pass(counter):
    d = tempfile.mkdtemp(prefix="bv2pmc_", dir="/tmp")
    cmd = ["rocprofv3", "--kernel-trace", "--pmc", counter, "-d", d, "--output-format", "csv", "--", sys.executable, "-c", SCRIPT]
    subprocess.run(cmd, cwd="/tmp", env=dict(os.environ, TMPDIR="/tmp"), stdout=subprocess.DEVNULL, stderr=subprocess.DEVNULL, check=False, timeout=900)
    tot, n = collections.Counter(), collections.Counter()
    seen = collections.defaultdict(set)
    for cc in glob.glob(d + "/**/*_counter_collection.csv", recursive=True):
        for r in csv.DictReader(open(cc)):
            if r["Counter_Name"] != counter:
                continue
            k = r["Kernel_Name"].split("(")[0]
            tot[k] += float(r["Counter_Value"])
            seen[k].add(r["Dispatch_Id"])
    for k, v in seen.items():
        n[k] = len(v)
    return tot, n


def main():
    out = sys.argv[1]
    fetch, nf = one_pass("FETCH_SIZE")
    write, nw = one_pass("WRITE_SIZE")
    fwd = max((c for k, c in nf.items() if "bert_embed_ln" in k), default=0)
    fwd_w = max((c for k, c in nw.items() if "bert_embed_ln" in k), default=0)
    kdir = os.path.join(ROOT, "bert-vits2_amd", "csrc", "kernels")
    res = {"_method": __doc__.split("    python")[0].strip(), "forwards_in_fetch_pass": fwd, "forwards_in_write_pass": fwd_w,
           "source_digests": {f: hashlib.sha256(open(os.path.join(kdir, f), "rb").read()).hexdigest()[:16] for f in SOURCES}, "kernels": {}}
    fb = wb = 0.0
    for k in sorted(set(fetch) | set(write)):
        if "bv2::" not in k:
            continue
        f = 2 * fetch.get(k, 0.0) * 1024 / max(fwd, 1)
        w = write.get(k, 0.0) * 1024 / max(fwd_w, 1)
        res["kernels"][k] = dict(launches_per_forward=round(nf.get(k, 0) / max(fwd, 1), 2), fetch_bytes_per_forward=f, write_bytes_per_forward=w)
        fb += f
        wb += w
    res["fetch_bytes_per_forward"] = fb
    res["write_bytes_per_forward"] = wb
    res["traffic_bytes_per_forward"] = fb + wb
    json.dump(res, open(out, "w"), indent=1)
    print(json.dumps({k: v for k, v in res.items() if k != "kernels"}, indent=1))
    for k, v in res["kernels"].items():
        print(f"{k[:90]:90s} {v['launches_per_forward']:6.1f} launches  fetch {v['fetch_bytes_per_forward'] / 1e6:9.2f} MB  write {v['write_bytes_per_forward'] / 1e6:8.2f} MB")


if __name__ == "__main__":
    main()
