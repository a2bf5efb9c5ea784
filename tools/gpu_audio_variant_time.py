"""A/B timing of the audio front-end across libamuse_hip*.so variants on ONE box (each variant in its own process, AMUSE_HIP_LIB;
tools/build_variant.sh NAME k_audio.hip "-D.. / flags").  Prints ms per encoder pass over B clips.  Usage: python tools/gpu_audio_variant_time.py [B]"""
import glob, os, subprocess, sys
REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if len(sys.argv) > 1 and sys.argv[1] == "--child":
    sys.path.insert(0, REPO)
    import torch
    from amuse_amd import audio_weights as aw
    from amuse_amd.audio import AudioEngine
    eng = AudioEngine(*(aw.make_ast_weights(0, n) for n in aw.ENCODERS))
    B = int(sys.argv[2])
    fb = eng.fbank(0.1 * torch.randn(B, 160000, device="cuda:0", generator=torch.Generator(device="cuda:0").manual_seed(1)))
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    ts = []
    for i in range(6):
        e0.record(); eng.encode(aw.ENCODERS[0], fb); e1.record(); torch.cuda.synchronize()
        if i >= 2:
            ts.append(e0.elapsed_time(e1))
    print(f"B={B}: one encoder min {min(ts):.3f} med {sorted(ts)[len(ts) // 2]:.3f} ms")
else:
    B = sys.argv[1] if len(sys.argv) > 1 else "32"
    libs = sorted(glob.glob(os.path.join(REPO, "amuse_amd", "libamuse_hip*.so")))
    for rnd in range(2):
        for lib in libs:
            r = subprocess.run([sys.executable, __file__, "--child", B], env=dict(os.environ, AMUSE_HIP_LIB=lib), capture_output=True, text=True)
            print(rnd, os.path.basename(lib).ljust(30), r.stdout.strip().splitlines()[-1] if r.stdout.strip() else r.stderr[-300:], flush=True)
