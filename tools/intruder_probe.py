"""Diagnostic: does a parked tenant (mstts_debug_park_cus on a side stream) keep the persistent decoder launch from becoming co-resident?"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests"))
import numpy as np, torch
from helpers import dims_pair, to_dev
from oracle import model as OM, train as OT
from multi_speaker_tts_amd.engine import TrainEngine
from multi_speaker_tts_amd import lib
dev = torch.device("cuda:0")
WIDE = dict(emb=64, enc_conv_ch=64, enc_lstm=256, spk=256, prenet=256, dec_lstm=1024, n_mel=16, post_ch=32)
pd, od = dims_pair(**WIDE)
eng = TrainEngine(pd, device=dev, values=OM.init_params(od, 3))
B, Te, L = 16, 64, 30
batch = to_dev(OT.synthetic_batch(od, B, Te, L, seed=5, ragged=True), dev)
w = eng.plan(B, Te, L)
eng.forward(batch, w, seed=11); torch.cuda.synchronize()
side = torch.cuda.Stream(device=dev)
for n_wg, park_us, sleep_ms in ((16, 30000, 3), (16, 30000, 0), (64, 30000, 3), (16, 6000, 0), (256, 30000, 3)):
    done = torch.zeros(1, dtype=torch.int32, device=dev)
    torch.cuda.synchronize()
    fb0 = eng.persist_fallbacks
    t0 = time.perf_counter()
    with torch.cuda.stream(side):
        lib.call("mstts_debug_park_cus", n_wg, park_us, lib.ptr(done))
    if sleep_ms:
        time.sleep(sleep_ms * 1e-3)
    t1 = time.perf_counter()
    eng.forward(batch, w, seed=11)
    torch.cuda.current_stream().synchronize()
    t2 = time.perf_counter()
    d_mid = int(done.item()) if False else -1
    torch.cuda.synchronize()
    t3 = time.perf_counter()
    print("park %3d wg x %5d us, host sleep %d ms: forward took %.2f ms, all done after %.2f ms, fallbacks +%d, status %r, parked done %d"
          % (n_wg, park_us, sleep_ms, (t2 - t1) * 1e3, (t3 - t0) * 1e3, eng.persist_fallbacks - fb0, getattr(eng, "persist_last_status", None), int(done.item())))
