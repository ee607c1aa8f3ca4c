"""Where a window strand's time goes on the stage-3 / stage-4 route (sv.Stage3Cache / Stage4Cache.encode): python tools/prof_stage3.py [n_variants] [level 3|4]
Per phase (HIP events, one context, stream order): MaxPool1d(5) gathers from the cache, the Encoder's front on the snippets (window ends, junctions),
stages 4-7 (`Encoder.back`).  Under `rocprofv3 --kernel-trace --stats` the kernel list of the same calls."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from orca_amd import engine, orca_models, sv
n = int(sys.argv[1]) if len(sys.argv) > 1 else 8
level = int(sys.argv[2]) if len(sys.argv) > 2 else 4
dev = torch.device("cuda:0")
h1 = orca_models.H1esc(synthetic_seed=0)
g = torch.Generator(device=dev).manual_seed(5)
genome = torch.randint(0, 4, (40_000_000,), device=dev, generator=g, dtype=torch.uint8)
svs = sv.synth_svs(n, 40_000_000)
s3 = (sv.Stage3Cache if level == 3 else sv.Stage4Cache)(h1.net0, genome)
torch.cuda.synchronize(); t = time.perf_counter()
assert s3.build_all()
torch.cuda.synchronize(); print(f"build: {time.perf_counter() - t:.2f} s")
acc = {"gather": 0.0, "snippets": 0.0, "back": 0.0, "n_snippets": 0, "snippet_bases": 0, "strands": 0}
ev = lambda: torch.cuda.Event(enable_timing=True)
out = torch.empty((128, 8000), device=dev)
with engine.defer_overflow_guard():
    for rep in range(2):       # the first pass warms every kernel variant up
        for v in svs:
            rp, rw, rm, ap, aw, am = sv.sv_windows(v, 40_000_000)
            for pieces in (rp, ap):
                w = sv.assemble_codes(genome, pieces)
                for rev in (False, True):
                    pcs = sv.revcomp_pieces(pieces) if rev else pieces
                    L, n4 = w.numel(), w.numel() // (5 * s3.grid)
                    kw = {} if level == 3 else dict(margin=sv.S4_MARGIN_BP, grid=sv.S4_GRID, pad=sv.S4_PAD_BP, min_snippet=sv.S4_MIN_SNIPPET_BP)
                    takes, snips = sv.s3_plan(pcs, s3.C, L, regions=s3.region, **kw)
                    s4 = torch.empty((32, engine.p16_plane_units(n4), 4) if level == 3 else (n4, 128), dtype=torch.float32, device=dev)
                    ctx = engine.get_context(dev)
                    e = [ev() for _ in range(4)]
                    e[0].record()
                    for m_lo, m_hi, _, strand, phase, c in takes:
                        (engine.p16_pool5_into if level == 3 else engine.rows_pool5_into)(ctx, s3.get(strand, phase), (c - s3._origin(strand, phase)) // s3.grid, s4, m_lo, m_hi - m_lo)
                    e[1].record()
                    if level == 4 and len(snips) > 1:      # as sv._s4_encode: the strand's snippets as ONE front run
                        L_ = w.numel()
                        cat = torch.cat([w[b0: b0 + nb] for _, _, b0, nb, _ in snips] if not rev else [w[L_ - b0 - nb: L_ - b0] for _, _, b0, nb, _ in reversed(snips)])
                        ranges, off = [], 0
                        for ga, gb, b0, nb, skip in snips:
                            ranges.append((off // 400 + skip, gb - ga, ga))
                            off += nb
                        h1.net0.front4_ranges(cat, rev, ranges, s4)
                    else:
                        for ga, gb, b0, nb, skip in snips:
                            (h1.net0.front_snippet if level == 3 else h1.net0.front4_snippet)(w, rev, b0, nb, skip, gb - ga, s4, ga)
                    e[2].record()
                    h1.net0.back(s4, n4, out) if level == 3 else h1.net0.back5(s4, out)
                    e[3].record()
                    torch.cuda.synchronize()
                    if rep:
                        acc["gather"] += e[0].elapsed_time(e[1]); acc["snippets"] += e[1].elapsed_time(e[2]); acc["back"] += e[2].elapsed_time(e[3])
                        acc["n_snippets"] += len(snips); acc["snippet_bases"] += sum(s_[3] for s_ in snips); acc["strands"] += 1
k = acc["strands"]
print({a: round(b / k, 3) for a, b in acc.items() if a != "strands"}, f"per window strand (ms; {k} strands)")
