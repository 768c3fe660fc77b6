#!/bin/bash
# Diagnostic build of the decoder kernels with the round-2 cross-attention variant and the device-side compare
# (tools/xattn_ab.py).  Output: tools/_build/libodic_dbg.so — never part of libodic_hip.so / the product.
set -e
cd "$(dirname "$0")/.."
mkdir -p tools/_build
/opt/rocm/bin/hipcc -O3 -std=c++17 -fPIC --offload-arch=gfx950 -ffp-contract=fast -Wall -Wno-unused-function \
  -DODIC_XATTN_VARIANTS -shared on_device_image_captioning_amd/csrc/decoder_ops.hip -o tools/_build/libodic_dbg.so
/opt/rocm/bin/hipcc -O3 -std=c++17 -fPIC --offload-arch=gfx950 -Wall -shared tools/csrc/pkfma_probe.hip -o tools/_build/libpkfma_probe.so
