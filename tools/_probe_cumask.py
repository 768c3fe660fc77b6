import ctypes, os, sys, time
sys.path.insert(0, os.getcwd())
import torch, bench
from on_device_image_captioning_amd import weights as W
from on_device_image_captioning_amd import pipeline as P
torch.set_grad_enabled(False)
hip = ctypes.CDLL("libamdhip64.so")
hip.hipExtStreamCreateWithCUMask.argtypes = [ctypes.POINTER(ctypes.c_void_p), ctypes.c_uint32, ctypes.POINTER(ctypes.c_uint32)]
hip.hipExtStreamCreateWithCUMask.restype = ctypes.c_int

def masked_stream(bits):
    words = (ctypes.c_uint32 * 8)(*[sum(1 << (b - 32 * w) for b in bits if 32 * w <= b < 32 * (w + 1)) for w in range(8)])
    s = ctypes.c_void_p()
    rc = hip.hipExtStreamCreateWithCUMask(ctypes.byref(s), 8, words)
    assert rc == 0, rc
    return torch.cuda.ExternalStream(s.value)

dev = torch.device("cuda", 0)
torch.cuda.set_device(0)
torch.zeros(1, device=dev)
model, sd, g = bench.build_model(dev, "bf16")
img = W.synth_images(16, g).to(dev)
ndec = int(sys.argv[1]) if len(sys.argv) > 1 else 32
cfgs = [("no mask", None), (f"decode {ndec} CUs", ndec)]
for name, nd in cfgs:
    streams = None
    if nd is not None:
        dec_bits = list(range(nd)); enc_bits = list(range(nd, 256))
        streams = (masked_stream(enc_bits), [masked_stream(dec_bits), masked_stream(dec_bits)])
    pipe = P.CaptionPipeline(model, 16, 3, 20, 79, 77, streams=streams)
    def both():
        pipe.submit(img)
        if pipe.full(): pipe.collect()
    for _ in range(5): both()
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(20): both()
    while pipe.outstanding(): pipe.collect()
    torch.cuda.synchronize()
    print(f"{name}: overlapped step {(time.perf_counter()-t0)/20*1e3:.3f} ms")
