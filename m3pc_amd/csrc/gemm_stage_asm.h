// One 64-deep K stage of the 256x256 (wave tile 128x64) bf16 GEMM, hand-scheduled: two fragment register
// sets; the 6 ds_read_b128 of k-step s+1 are in flight under the 8 MFMAs of k-step s; counted lgkmcnt.
// Operands: %0-%7 accumulators [i][j] (i<4, j<2), %8-%13 / %14-%19 fragment sets, %20-%23 A-row LDS
// addresses per k-step, %24-%27 W-row LDS addresses per k-step (rows 32 apart are offset:4096).
#pragma once
#define M3PC_STAGE_ASM_4x2 \
    "ds_read_b128 %8, %20\n\t" \
    "ds_read_b128 %9, %20 offset:4096\n\t" \
    "ds_read_b128 %10, %20 offset:8192\n\t" \
    "ds_read_b128 %11, %20 offset:12288\n\t" \
    "ds_read_b128 %12, %24\n\t" \
    "ds_read_b128 %13, %24 offset:4096\n\t" \
    "ds_read_b128 %14, %21\n\t" \
    "ds_read_b128 %15, %21 offset:4096\n\t" \
    "ds_read_b128 %16, %21 offset:8192\n\t" \
    "ds_read_b128 %17, %21 offset:12288\n\t" \
    "ds_read_b128 %18, %25\n\t" \
    "ds_read_b128 %19, %25 offset:4096\n\t" \
    "s_waitcnt lgkmcnt(6)\n\t" \
    "v_mfma_f32_32x32x16_bf16 %0, %8, %12, %0\n\t" \
    "v_mfma_f32_32x32x16_bf16 %1, %8, %13, %1\n\t" \
    "v_mfma_f32_32x32x16_bf16 %2, %9, %12, %2\n\t" \
    "v_mfma_f32_32x32x16_bf16 %3, %9, %13, %3\n\t" \
    "v_mfma_f32_32x32x16_bf16 %4, %10, %12, %4\n\t" \
    "v_mfma_f32_32x32x16_bf16 %5, %10, %13, %5\n\t" \
    "v_mfma_f32_32x32x16_bf16 %6, %11, %12, %6\n\t" \
    "v_mfma_f32_32x32x16_bf16 %7, %11, %13, %7\n\t" \
    "ds_read_b128 %8, %22\n\t" \
    "ds_read_b128 %9, %22 offset:4096\n\t" \
    "ds_read_b128 %10, %22 offset:8192\n\t" \
    "ds_read_b128 %11, %22 offset:12288\n\t" \
    "ds_read_b128 %12, %26\n\t" \
    "ds_read_b128 %13, %26 offset:4096\n\t" \
    "s_waitcnt lgkmcnt(6)\n\t" \
    "v_mfma_f32_32x32x16_bf16 %0, %14, %18, %0\n\t" \
    "v_mfma_f32_32x32x16_bf16 %1, %14, %19, %1\n\t" \
    "v_mfma_f32_32x32x16_bf16 %2, %15, %18, %2\n\t" \
    "v_mfma_f32_32x32x16_bf16 %3, %15, %19, %3\n\t" \
    "v_mfma_f32_32x32x16_bf16 %4, %16, %18, %4\n\t" \
    "v_mfma_f32_32x32x16_bf16 %5, %16, %19, %5\n\t" \
    "v_mfma_f32_32x32x16_bf16 %6, %17, %18, %6\n\t" \
    "v_mfma_f32_32x32x16_bf16 %7, %17, %19, %7\n\t" \
    "ds_read_b128 %14, %23\n\t" \
    "ds_read_b128 %15, %23 offset:4096\n\t" \
    "ds_read_b128 %16, %23 offset:8192\n\t" \
    "ds_read_b128 %17, %23 offset:12288\n\t" \
    "ds_read_b128 %18, %27\n\t" \
    "ds_read_b128 %19, %27 offset:4096\n\t" \
    "s_waitcnt lgkmcnt(6)\n\t" \
    "v_mfma_f32_32x32x16_bf16 %0, %8, %12, %0\n\t" \
    "v_mfma_f32_32x32x16_bf16 %1, %8, %13, %1\n\t" \
    "v_mfma_f32_32x32x16_bf16 %2, %9, %12, %2\n\t" \
    "v_mfma_f32_32x32x16_bf16 %3, %9, %13, %3\n\t" \
    "v_mfma_f32_32x32x16_bf16 %4, %10, %12, %4\n\t" \
    "v_mfma_f32_32x32x16_bf16 %5, %10, %13, %5\n\t" \
    "v_mfma_f32_32x32x16_bf16 %6, %11, %12, %6\n\t" \
    "v_mfma_f32_32x32x16_bf16 %7, %11, %13, %7\n\t" \
    "s_waitcnt lgkmcnt(0)\n\t" \
    "v_mfma_f32_32x32x16_bf16 %0, %14, %18, %0\n\t" \
    "v_mfma_f32_32x32x16_bf16 %1, %14, %19, %1\n\t" \
    "v_mfma_f32_32x32x16_bf16 %2, %15, %18, %2\n\t" \
    "v_mfma_f32_32x32x16_bf16 %3, %15, %19, %3\n\t" \
    "v_mfma_f32_32x32x16_bf16 %4, %16, %18, %4\n\t" \
    "v_mfma_f32_32x32x16_bf16 %5, %16, %19, %5\n\t" \
    "v_mfma_f32_32x32x16_bf16 %6, %17, %18, %6\n\t" \
    "v_mfma_f32_32x32x16_bf16 %7, %17, %19, %7\n\t"
