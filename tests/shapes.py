"""Shape tables shared by the kernel parity tests (GPU) and the variant-coverage test (CPU)."""

# (M, N, K) of usf_linear_f32 launches with bf16x3 planes that tests/test_kernels_gpu.py compares with fp64 / fp32
# torch-CPU arithmetic
BF16X3_SMALL = [(65, 65, 8), (200, 160, 40), (257, 784, 784), (1000, 392, 256), (4096, 800, 784),
                # the 256 x 160 tile with its ring of four weight buffers: 1, 2, 3, 4 slabs, ragged rows/cols
                (8205, 800, 8), (8205, 800, 40), (8205, 800, 72), (8205, 800, 104), (8205, 784, 776),
                # the 128 x 128 tile
                (8192, 1024, 64), (8199, 1020, 136)]
# BASELINE shapes at full size (reference arithmetic on head / middle / tail rows of the batch)
BF16X3_BIG = [(65536, 784, 784),      # cfg2 affine
              (32768, 784, 784),      # cfg3 per rank
              (16384, 784, 784),
              (125000, 784, 784),     # cfg5 per rank
              (32768, 3072, 3072),    # cfg4 affine
              (32768, 1024, 1536),    # cfg4 conditioner: first layer (mask-aware K = D/2)
              (32768, 1024, 1024),    # ... hidden layer
              (32768, 1536, 1024),    # ... output layer
              (65536, 256, 392), (65536, 256, 256), (65536, 392, 256)]   # cfg2 conditioner as a chain of linears

# every usf_linear_f32 launch shape (M, N, K) of the BASELINE configurations' launch plans in bf16x3 mode
BASELINE_LINEAR_SHAPES = {
    "cfg2 (B=65536)": [(65536, 784, 784), (65536, 256, 392), (65536, 256, 256), (65536, 392, 256)],
    "cfg3 per rank (B=32768)": [(32768, 784, 784), (32768, 256, 392), (32768, 256, 256), (32768, 392, 256)],
    "cfg4 (B=32768)": [(32768, 3072, 3072), (32768, 1024, 1536), (32768, 1024, 1024), (32768, 1536, 1024)],
    "cfg5 per rank (N=125000)": [(125000, 784, 784), (125000, 256, 392), (125000, 256, 256), (125000, 392, 256)],
}

# planes pipeline: (M, K blocks, output blocks or fp32 N as a negative number) of the usf_gemm_planes_bf16x3 launches that
# tests/test_planes_gpu.py compares with reference arithmetic ...
PLANES_TESTED = [(100, 3, 2), (1000, 25, 25), (3000, 13, 8), (3000, 8, 13), (8205, 25, -784), (513, 2, -77),
                 (65536, 25, 25), (32768, 96, 96)]
# ... and the launches of the BASELINE configurations' plans
BASELINE_PLANES_SHAPES = {
    "cfg2 / cfg3 / cfg5": [(65536, 25, 25), (65536, 13, 8), (65536, 8, 8), (65536, 8, 13), (65536, 25, -784)],
    "cfg4": [(32768, 96, 96), (32768, 48, 32), (32768, 32, 32), (32768, 32, 48), (32768, 96, -3072)],
}
