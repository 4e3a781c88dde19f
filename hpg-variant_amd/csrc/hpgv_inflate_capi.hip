// hpgv_inflate_capi.hip -- C ABI of the bgzip decoder (its own translation unit of libhpgv.so: the wave-per-block kernel has
// wave-uniform branches only and is compiled with -mllvm -structurizecfg-skip-uniform-regions, which leaves them as written).
#include "hpgv_internal.h"
#ifdef HPGV_ABLATION
#include "hpgv_inflate_kernels.h"      // one lane per block (two forms): lost to the wave-per-block decoder
#endif
#include "hpgv_inflate2_kernels.h"
#include "hpgv_bgzf_kernels.h"
#include "hpgv_crc_kernels.h"

extern "C" {

// raw-DEFLATE blocks (the payloads of BGZF blocks) -> text, all on the device: block b occupies d_comp[in_off[b] .. +in_len[b])
// and decodes to exactly out_len[b] bytes at d_text + out_off[b]; d_status[b] = 0, or a non-zero code for a block this decoder
// does not take (the host then decodes that block).  One wave per block, several symbols per round of its loop.  (An ablation
// build, -DHPGV_ABLATION, also holds the forms it beat -- option inflate_wave: 2 = one symbol per round; 0 = one lane per block
// (wants a hundred thousand blocks per call), 3 = the lane kernel with its symbol tables in LDS.)
int hpgv_inflate_blocks_dev(hpgv_ctx *ctx, const uint8_t *d_comp, const uint64_t *d_in_off, const uint32_t *d_in_len,
                            const uint64_t *d_out_off, const uint32_t *d_out_len, int n_blocks, uint8_t *d_text,
                            int32_t *d_status, void *stream) {
    ctx = first_member(ctx);
    if (!ctx) return HPGV_ERR_INVALID;
    if (n_blocks < 0 || (n_blocks > 0 && (!d_comp || !d_in_off || !d_in_len || !d_out_off || !d_out_len || !d_text || !d_status)))
        return fail(ctx, HPGV_ERR_INVALID, "bad inflate arguments");
    if (n_blocks == 0) return HPGV_OK;
    DeviceGuard g(ctx->device);
    // one wave per block: a block in a millisecond whatever the number of blocks (4 096 blocks: 1.3 ms), 390 GB/s from
    // 125 000 blocks on (one symbol per round: 262 - 267), and only the job's own bytes move; one lane per block
    // (inflate_wave = 0): 13 - 38 ms for a launch of any size, 230 - 258 GB/s from 125 000 blocks on, ten times the job's bytes
    // through HBM.  inflate_wave = 1 is "the library's choice": the wave kernel with several symbols per round
#ifndef HPGV_ABLATION
    // the shipped decoder: a wave per block, several symbols per round of its loop (hpgv_inflate2_kernels.h)
    hipLaunchKernelGGL(hpgv::k_inflate_wave<true>, dim3((unsigned)n_blocks), dim3(64), 0, (hipStream_t)stream,
                       d_comp, d_in_off, d_in_len, d_out_off, d_out_len, n_blocks, d_text, d_status);
#else
    const long mode = ctx->inflate_wave;
    const bool wave = mode == 2 || mode == 1 || mode == 4;
    // (experiment: unused dynamic LDS per wave caps the waves per compute unit and leaves LDS for the kernels beside it)
    const unsigned lds_pad = (unsigned)ctx->inflate_lds_pad;
    if (wave) {
        // (experiment: inflate_wave_wgs = waves per compute unit in flight, each going on to further blocks; 0 = a wave per block)
        const unsigned per_cu = (unsigned)ctx->inflate_wave_wgs;
        unsigned grid = (unsigned)n_blocks;
        if (per_cu && grid > per_cu * (unsigned)ctx->n_cus) grid = per_cu * (unsigned)ctx->n_cus;
        if (mode != 2)                                               // several symbols per round (hpgv_inflate2_kernels.h); 2: one symbol per round (A/B)
            hipLaunchKernelGGL(hpgv::k_inflate_wave<true>, dim3(grid), dim3(64), lds_pad, (hipStream_t)stream,
                               d_comp, d_in_off, d_in_len, d_out_off, d_out_len, n_blocks, d_text, d_status);
        else
            hipLaunchKernelGGL(hpgv::k_inflate_wave<false>, dim3(grid), dim3(64), lds_pad, (hipStream_t)stream,
                               d_comp, d_in_off, d_in_len, d_out_off, d_out_len, n_blocks, d_text, d_status);
    }
    else if (mode != 3) {
        // (experiment: inflate_lane_wgs = workgroups per compute unit in flight; 0 = one per 64 blocks)
        const unsigned per_cu = (unsigned)ctx->inflate_lane_wgs;
        unsigned grid = (unsigned)((n_blocks + 63) / 64);
        if (per_cu && grid > per_cu * (unsigned)ctx->n_cus) grid = per_cu * (unsigned)ctx->n_cus;
        hipLaunchKernelGGL(hpgv::k_inflate_blocks, dim3(grid), dim3(64), 0, (hipStream_t)stream,
                           d_comp, d_in_off, d_in_len, d_out_off, d_out_len, n_blocks, d_text, d_status);
    } else                                                           // (A/B: the lane kernel with its symbol tables in LDS; profiles/experiments_that_did_not_pay.md)
        hipLaunchKernelGGL(hpgv::k_inflate_blocks_lds, dim3((unsigned)((n_blocks + 63) / 64)), dim3(64), 0, (hipStream_t)stream,
                           d_comp, d_in_off, d_in_len, d_out_off, d_out_len, n_blocks, d_text, d_status);
#endif
    HIPCHK(ctx, hipGetLastError());
    return HPGV_OK;
}

// The CRC-32 check of decoded BGZF blocks (htslib's bgzf reader and zlib's gzread reject a block whose text does not have the
// CRC its trailer gives; a DEFLATE stream can be damaged and still inflate to ISIZE bytes).  For every block whose d_status is
// 0: CRC-32 of its out_len bytes at d_text + out_off against the four bytes at d_comp + in_off + in_len (the trailer follows
// the payload); a mismatch sets d_status to HPGV_BLOCK_BAD_CRC.  Asynchronous on `stream`, behind the decoder's launch.
int hpgv_bgzf_verify_dev(hpgv_ctx *ctx, const uint8_t *d_comp, const uint64_t *d_in_off, const uint32_t *d_in_len,
                         const uint64_t *d_out_off, const uint32_t *d_out_len, int n_blocks, const uint8_t *d_text,
                         int32_t *d_status, void *stream) {
    return hpgv_bgzf_verify_tiles_dev(ctx, d_comp, d_in_off, d_in_len, d_out_off, d_out_len, n_blocks, d_text, d_status, nullptr, 0, stream);
}

size_t hpgv_text_tiles_bytes(uint64_t text_bytes) { return (size_t)((text_bytes + hpgv::TOK2_TILE - 1) / hpgv::TOK2_TILE + 1) * sizeof(hpgv::TokAgg2); }

// ... and, with d_tiles, the tokenizer's records of the decoded text (hpgv.h "hpgv_text_alias_tiles"): d_tiles holds
// hpgv_text_tiles_bytes(text) bytes, ZEROED before the first block of the text is verified; n_tiles of them may be written
int hpgv_bgzf_verify_tiles_dev(hpgv_ctx *ctx, const uint8_t *d_comp, const uint64_t *d_in_off, const uint32_t *d_in_len,
                               const uint64_t *d_out_off, const uint32_t *d_out_len, int n_blocks, const uint8_t *d_text,
                               int32_t *d_status, void *d_tiles, uint64_t n_tiles, void *stream) {
    HPGV_ABI_TRY
    ctx = first_member(ctx);
    if (!ctx) return HPGV_ERR_INVALID;
    if (n_blocks < 0 || (n_blocks > 0 && (!d_comp || !d_in_off || !d_in_len || !d_out_off || !d_out_len || !d_text || !d_status)))
        return fail(ctx, HPGV_ERR_INVALID, "bad block check arguments");
    if (n_blocks == 0) return HPGV_OK;
    DeviceGuard g(ctx->device);
    {
        std::lock_guard<std::mutex> lk(ctx->mu);
        if (!ctx->d_crc_tab) {
            static_assert(HPGV_BLOCK_BAD_CRC == hpgv::BGZF_STATUS_BAD_CRC, "status code of the header and of the kernel");
            std::vector<uint32_t> tab(hpgv::CRC_TAB_WORDS);
            hpgv::crc_build_tables(tab.data());
            HIPCHK(ctx, hipMalloc(&ctx->d_crc_tab, tab.size() * sizeof(uint32_t)));
            HIPCHK(ctx, hipMemcpy(ctx->d_crc_tab, tab.data(), tab.size() * sizeof(uint32_t), hipMemcpyHostToDevice));
        }
    }
    if (d_tiles && n_tiles > 0)
        hipLaunchKernelGGL(hpgv::k_bgzf_crc<true>, dim3((unsigned)((n_blocks + 3) / 4)), dim3(256), 0, (hipStream_t)stream, d_comp, d_in_off, d_in_len,
                           d_out_off, d_out_len, n_blocks, d_text, d_status, (const uint32_t *)ctx->d_crc_tab, (hpgv::TokAgg2 *)d_tiles, (long)n_tiles);
    else
        hipLaunchKernelGGL(hpgv::k_bgzf_crc<false>, dim3((unsigned)((n_blocks + 3) / 4)), dim3(256), 0, (hipStream_t)stream, d_comp, d_in_off, d_in_len,
                           d_out_off, d_out_len, n_blocks, d_text, d_status, (const uint32_t *)ctx->d_crc_tab, (hpgv::TokAgg2 *)nullptr, 0L);
    HIPCHK(ctx, hipGetLastError());
    return HPGV_OK;
    HPGV_ABI_CATCH(ctx)
}

// The block table of a bgzip file from its compressed bytes on the device: the blocks that form a chain from byte `lo`
// (a block start) and end at or before `hi` (bytes [0, hi) are there), at most max_rows of them, as rows of the decoder's
// tables (d_in_off .. d_out_len, index 0 on; out_off counts from text_base).  result[0] = rows, [1] = where the chain
// stands now (the next call's lo), [2] = text_base + the rows' text bytes, [3] = headers seen in the range.  Zero rows with
// bytes to spare means the file's headers are not the ones bgzip writes: walk it on the host.  The call returns when the
// result is there; d_scratch holds hpgv_bgzf_scan_scratch_bytes(hi - lo, max_rows) bytes.
size_t hpgv_bgzf_scan_scratch_bytes(uint64_t range_bytes, int max_rows) {
    const size_t tiles = (size_t)((range_bytes + 15 + hpgv::BGZF_TILE - 1) / hpgv::BGZF_TILE) + 1;
    return 64 + ((tiles * 4 + 63) & ~(size_t)63) + ((size_t)(max_rows > 0 ? max_rows : 0) + 64) * sizeof(hpgv::BgzfHit);
}
int hpgv_bgzf_scan_dev(hpgv_ctx *ctx, const uint8_t *d_comp, uint64_t lo, uint64_t hi, uint64_t text_base, int max_rows,
                       uint64_t *d_in_off, uint32_t *d_in_len, uint64_t *d_out_off, uint32_t *d_out_len,
                       void *d_scratch, size_t scratch_bytes, uint64_t *result, void *stream) {
    HPGV_ABI_TRY
    ctx = first_member(ctx);
    if (!ctx) return HPGV_ERR_INVALID;
    if (!d_comp || hi < lo || max_rows <= 0 || !d_in_off || !d_in_len || !d_out_off || !d_out_len || !d_scratch || !result ||
        scratch_bytes < hpgv_bgzf_scan_scratch_bytes(hi - lo, max_rows) || hi - lo > ((uint64_t)1 << 40))
        return fail(ctx, HPGV_ERR_INVALID, "bad block scan arguments");
    DeviceGuard g(ctx->device);
    hipStream_t st = (hipStream_t)stream;
    const uint64_t base = lo & ~(uint64_t)15;
    const int n_tiles = (int)((hi - base + hpgv::BGZF_TILE - 1) / hpgv::BGZF_TILE);
    uint64_t *d_result = (uint64_t *)d_scratch;                     // 4 x u64, then the total, then the tiles, then the hits
    uint32_t *d_total = (uint32_t *)((char *)d_scratch + 32), *d_tiles = (uint32_t *)((char *)d_scratch + 64);
    const size_t tiles_bytes = (((size_t)n_tiles + 1) * 4 + 63) & ~(size_t)63;
    hpgv::BgzfHit *d_hit = (hpgv::BgzfHit *)((char *)d_scratch + 64 + tiles_bytes);
    const uint32_t cap = (uint32_t)max_rows + 64;
    if (n_tiles > 0) {
        hipLaunchKernelGGL(hpgv::k_bgzf_count, dim3((unsigned)n_tiles), dim3(hpgv::BGZF_TPB), 0, st, d_comp, base, lo, hi, d_tiles);
        hipLaunchKernelGGL(hpgv::k_bgzf_scan, dim3(1), dim3(1024), 0, st, d_tiles, n_tiles, d_total);
        hipLaunchKernelGGL(hpgv::k_bgzf_list, dim3((unsigned)n_tiles), dim3(hpgv::BGZF_TPB), 0, st, d_comp, base, lo, hi, (const uint32_t *)d_tiles, d_hit, cap);
    } else {
        HIPCHK(ctx, hipMemsetAsync(d_total, 0, 4, st));
    }
    hipLaunchKernelGGL(hpgv::k_bgzf_chain, dim3(1), dim3(1024), 0, st, (const hpgv::BgzfHit *)d_hit, (const uint32_t *)d_total, cap, lo, text_base,
                       (uint32_t)max_rows, d_in_off, d_in_len, d_out_off, d_out_len, d_result);
    HIPCHK(ctx, hipGetLastError());
    HIPCHK(ctx, hipMemcpyAsync(result, d_result, 32, hipMemcpyDeviceToHost, st));
    HIPCHK(ctx, hipStreamSynchronize(st));
    return HPGV_OK;
    HPGV_ABI_CATCH(ctx)
}

}  // extern "C"
