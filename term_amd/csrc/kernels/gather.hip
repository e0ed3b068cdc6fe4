// gather.hip -- small RecordBatches made one: the segment gather of the library-side batch coalescing (gfx950).
//
// DataFusion hands out 8192-row batches (TG/core/context.rs:28-38, `batch_size: 8192`): a launch (let alone a
// host-device synchronisation) per batch starves every kernel of the path.  tgx_update therefore only NOTES small
// batches -- one descriptor per (column, batch): where the window lives (the caller's DEVICE buffer, or the pinned
// arena a HOST batch was copied into) -- and a flush turns the pending segments of every column into ONE contiguous
// Arrow column on the device, which then goes through the same fused pass as a big batch:
//   * fixed-width values (8 / 4 bytes): copied window by window (16-byte moves where both sides allow),
//   * validity bitmaps: concatenated bit-exactly -- a segment may start at any Arrow offset and land at any row, so
//     every destination word is assembled from the source bits it covers; words shared by two segments are merged
//     with an atomic OR into the zeroed destination; a segment without a bitmap contributes ones,
//   * Utf8 / LargeUtf8: offsets re-based onto the running byte position of the coalesced data buffer, bytes copied,
//   * Utf8View (what DataFusion reads Parquet strings as): the stretches of the data buffers a window's long views
//     point into are laid one behind the other in ONE data buffer and the views re-pointed at them,
//   * Dictionary<Int32, Utf8>: the windows' dictionaries (one per run of batches that share theirs) are gathered like
//     a Utf8 column of their own and every window's indices shifted to where its dictionary starts.
// One workgroup per (column, segment) -- several for windows of more than ~8192 rows.  HBM traffic: the windows are read once and written once (then read by the
// checks): a stream of 8192-row batches costs 3x the bytes of one big batch -- against one launch per ~500 batches.
#include <hip/hip_runtime.h>

#include "device_types.h"

namespace tgx {

typedef const uint8_t __attribute__((address_space(1))) *g_u8;

// (tid, nt): this thread's index among the nt threads that share the move -- the workgroups blockIdx.y = 0 .. gridDim.y - 1
// of a segment (a 64 Ki-row window moved by ONE workgroup took 0.8 ms: a flush of fifteen such windows kept 15 CUs busy)
__device__ __forceinline__ void gather_bytes(uint8_t *__restrict__ dst, const uint8_t *__restrict__ src, int64_t bytes,
                                             const int tid, const int nt) {
  const uintptr_t mis = ((uintptr_t)dst | (uintptr_t)src);
  if ((mis & 15) == 0) {
    const int64_t n16 = bytes >> 4;
    typedef unsigned int u32x4 __attribute__((ext_vector_type(4)));
    const u32x4 *s = (const u32x4 *)src;
    u32x4 *d = (u32x4 *)dst;
    // four moves in flight per lane
    int64_t i = tid;
    for (; i + 3 * (int64_t)nt < n16; i += 4 * (int64_t)nt) {
      const u32x4 a = __builtin_nontemporal_load(s + i), b = __builtin_nontemporal_load(s + i + nt);
      const u32x4 c = __builtin_nontemporal_load(s + i + 2 * nt), e = __builtin_nontemporal_load(s + i + 3 * nt);
      d[i] = a;
      d[i + nt] = b;
      d[i + 2 * nt] = c;
      d[i + 3 * nt] = e;
    }
    for (; i < n16; i += nt) d[i] = s[i];
    for (int64_t k = (n16 << 4) + tid; k < bytes; k += nt) dst[k] = src[k];
  } else if ((mis & 7) == 0) {
    const int64_t n8 = bytes >> 3;
    const uint64_t *s = (const uint64_t *)src;
    uint64_t *d = (uint64_t *)dst;
    for (int64_t i = tid; i < n8; i += nt) d[i] = s[i];
    for (int64_t k = (n8 << 3) + tid; k < bytes; k += nt) dst[k] = src[k];
  } else if ((mis & 3) == 0) {
    const int64_t n4 = bytes >> 2;
    const uint32_t *s = (const uint32_t *)src;
    uint32_t *d = (uint32_t *)dst;
    for (int64_t i = tid; i < n4; i += nt) d[i] = s[i];
    for (int64_t k = (n4 << 2) + tid; k < bytes; k += nt) dst[k] = src[k];
  } else {
    // (string bytes: source and destination positions are unrelated) -- destination words assembled from source bytes
    const int64_t head = bytes < 4 ? bytes : (int64_t)((4 - ((uintptr_t)dst & 3)) & 3);
    for (int64_t k = tid; k < head; k += nt) dst[k] = src[k];
    const int64_t n4 = (bytes - head) >> 2;
    uint32_t *d = (uint32_t *)(dst + head);
    const uint8_t *s = src + head;
    for (int64_t i = tid; i < n4; i += nt) {
      const uint8_t *q = s + 4 * i;
      d[i] = (uint32_t)q[0] | ((uint32_t)q[1] << 8) | ((uint32_t)q[2] << 16) | ((uint32_t)q[3] << 24);
    }
    for (int64_t k = head + (n4 << 2) + tid; k < bytes; k += nt) dst[k] = src[k];
  }
}

__global__ __launch_bounds__(256) void gather_segments_kernel(const GatherSeg *__restrict__ segs) {
  // (by reference: the descriptor's arrays are indexed at run time in the Utf8View branch, and a copy of it would live
  // in scratch memory -- 264 bytes written and read by every thread of every window, as much as a window's values)
  const GatherSeg &g = segs[blockIdx.x];
  const int tid = (int)(blockIdx.y * blockDim.x + threadIdx.x), nt = (int)(gridDim.y * blockDim.x);
  if (g.length <= 0) return;
  // ---- validity: destination words [w0, w1] of this segment's rows ----
  if (g.dst_validity) {
    uint32_t *dv = (uint32_t *)g.dst_validity;
    const int64_t r0 = g.dst_row, r1 = g.dst_row + g.length;
    const int64_t w0 = r0 >> 5, w1 = (r1 - 1) >> 5;
    for (int64_t w = w0 + tid; w <= w1; w += nt) {
      const int64_t lo = (w << 5) > r0 ? (w << 5) : r0;
      const int64_t hi = ((w + 1) << 5) < r1 ? ((w + 1) << 5) : r1;
      const uint32_t nb = (uint32_t)(hi - lo);
      const uint32_t mask = nb >= 32 ? 0xFFFFFFFFu : ((1u << nb) - 1u);
      uint32_t bits = mask;
      if (g.src_validity) {
        const int64_t sb = g.src_bit0 + (lo - r0);
        const uint8_t *q = g.src_validity + (sb >> 3);
        const uint32_t sh = (uint32_t)(sb & 7);
        const uint32_t need = (sh + nb + 7) >> 3;  // <= 5 source bytes; nothing past the window's last byte is read
        uint64_t v = 0;
        for (uint32_t k = 0; k < need; k++) v |= (uint64_t)q[k] << (8 * k);
        bits = (uint32_t)(v >> sh) & mask;
      }
      const uint32_t word = bits << (uint32_t)(lo - (w << 5));
      if (nb == 32)
        dv[w] = word;
      else if (word)
        atomicOr(dv + w, word);  // a word shared with the neighbouring segment (the destination was zeroed)
    }
  }
  // ---- values ----
  if (g.kind == 0) {
    if (g.src_values)
      gather_bytes((uint8_t *)g.dst_values + g.dst_row * g.elem_bytes, (const uint8_t *)g.src_values,
                   g.length * g.elem_bytes, tid, nt);
    return;
  }
  if (g.kind == 4) {  // dictionary indices: the window's dictionary starts at index_shift of the coalesced one
    const int32_t *si = (const int32_t *)g.src_values;
    int32_t *d = (int32_t *)g.dst_values + g.dst_row;
    for (int64_t i = tid; i < g.length; i += nt) d[i] = si[i] + g.index_shift;
    return;
  }
  if (g.kind == 3) {  // Utf8View: the stretches of the data buffers, then the views re-pointed at them
    for (int k = 0; k < g.vb_count; k++)
      if (g.vb_len[k] > 0) gather_bytes(g.dst_data + g.vb_base[k], g.vb_src[k], g.vb_len[k], tid, nt);
    const uint4 *sv = (const uint4 *)g.src_values;
    uint4 *d = (uint4 *)g.dst_values + g.dst_row;
    for (int64_t i = tid; i < g.length; i += nt) {
      uint4 v = sv[i];
      if ((int32_t)v.x > 12) {
        int k = 0;
        while (k < g.vb_count && g.vb_index[k] != (int32_t)v.z) k++;
        if (k < g.vb_count) {
          v.w = (uint32_t)(g.vb_base[k] + ((int64_t)(int32_t)v.w - g.vb_min[k]));
          v.z = 0;
        } else {
          v = make_uint4(0, 0, 0, 0);  // (a NULL row's view may hold anything: it becomes an empty value)
        }
      } else if ((int32_t)v.x < 0) {
        v = make_uint4(0, 0, 0, 0);
      }
      d[i] = v;
    }
    return;
  }
  // ---- strings: offsets re-based, bytes copied ----
  const int64_t shift = g.data_base - g.data_first;
  if (g.kind == 1) {
    const int32_t *so = (const int32_t *)g.src_values;
    int32_t *d = (int32_t *)g.dst_values + g.dst_row;
    for (int64_t i = tid; i <= g.length; i += nt) d[i] = (int32_t)((int64_t)so[i] + shift);
  } else {
    const int64_t *so = (const int64_t *)g.src_values;
    int64_t *d = (int64_t *)g.dst_values + g.dst_row;
    for (int64_t i = tid; i <= g.length; i += nt) d[i] = so[i] + shift;
  }
  if (g.data_len > 0) gather_bytes(g.dst_data + g.data_base, g.src_data, g.data_len, tid, nt);
}

// `parts`: workgroups per segment (1 for DataFusion's 8192-row windows; a 64 Ki-row window is shared by eight)
void launch_gather_segments(const GatherSeg *d_segs, int n_segs, int parts, hipStream_t stream) {
  if (n_segs <= 0) return;
  if (parts < 1) parts = 1;
  hipLaunchKernelGGL(gather_segments_kernel, dim3(n_segs, parts), dim3(256), 0, stream, d_segs);
}

}  // namespace tgx
