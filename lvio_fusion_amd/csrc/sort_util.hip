// sort_util.hip — a STABLE device sort of (32-bit key, index) pairs: hand-written LSD radix sort for gfx950, digits of up to 11 bits.
//
// Used where the order inside equal keys has to be the INPUT order: the voxel grid accumulates each voxel's centroid in float over its
// points in ascending input index (cloud_kernels.hip), which is what makes the filter's output independent of scheduling and equal to the
// sequential restatement bit for bit.
//
// Per digit pass, three launches on the context's stream and no host wait:
//   k_radix_hist     one workgroup per tile of 4 096 keys: histogram of the digit (<= 2 048 bins) in LDS -> hist[bin][tile] (digit-major, so that ONE exclusive
//                    scan over the whole table yields every (bin, tile)'s first output slot);
//   k_radix_scan     that scan, one workgroup of 1 024 threads (the table has bins x tiles entries: 25 k for a 100 k-point cloud and 10-bit digits;
//                    tables beyond 256 k entries — clouds of millions of points — take k_radix_scan_sums + k_radix_scan_chunks, up to 1 024 workgroups);
//   k_radix_scatter  the tile again, four rounds of 1 024 keys in input order: a key's rank among the keys of its digit is
//                    (keys of that digit in earlier rounds of the tile) + (in lower waves of this round) + (in lower lanes of its wave) —
//                    the last from one wave ballot per digit bit (lanes whose digit equals mine), no sorting network, no atomics,
//                    so equal digits keep their input order: stable.
// Keys of at most `key_bits` bits take ceil(key_bits / 11) passes of equal digit width (20-bit voxel keys: two 10-bit passes), ping-ponging between the output arrays and a pooled scratch pair.
// (Round 3 called hipcub::DeviceRadixSort here and waited for the stream after it.)
#include <algorithm>
#include "lvf_internal.hpp"

namespace lvf {

// (tiles of 4 096 keys: a quarter of the histogram table of 1 024-key tiles — the one-workgroup scan over it was 29 of a pass's 41 us)
constexpr int kRT = 256, kRRounds = 16, kRTile = kRT * kRRounds, kRScanT = 1024, kRMaxDigitBits = kSortMaxDigitBits;

// digit = (key >> shift) & (nbins - 1); nbins = 1 << digit_bits <= 2048.  Dynamic LDS: nbins ints.
// `mask`: the digit's live bits — nbins - 1, except in the LAST pass of a key whose width is not a multiple of the digit width, where only the
// remaining key bits count (bits above key_bits never influence the order: the contract hipcub's end_bit gave the callers)
__device__ __forceinline__ void radix_hist_body(int n, const unsigned* __restrict__ keys, int shift, int nbins, unsigned mask, int* __restrict__ hist, int ntiles, int* rs_lds) {
  int* h = rs_lds;
  for (int b = threadIdx.x; b < nbins; b += kRT) h[b] = 0;
  __syncthreads();
  const int base = blockIdx.x * kRTile;
  if (base < n) {                          // (uniform; a tile beyond the keys only writes its zeros)
    // the thread's sixteen keys are requested together (index clamped), then counted: as `if (i < n) atomicAdd(&h[key(keys[i])], 1)` every
    // round waited for its own load before its LDS atomic — sixteen dependent round trips
    unsigned kk[kRRounds];
#pragma unroll
    for (int r = 0; r < kRRounds; ++r) kk[r] = keys[min(base + r * kRT + (int)threadIdx.x, n - 1)];
#pragma unroll
    for (int r = 0; r < kRRounds; ++r) if (base + r * kRT + (int)threadIdx.x < n) atomicAdd(&h[(kk[r] >> shift) & mask], 1);
  }
  __syncthreads();
  for (int b = threadIdx.x; b < nbins; b += kRT) hist[(size_t)b * ntiles + blockIdx.x] = h[b];
}
__global__ __launch_bounds__(kRT) void k_radix_hist(int n, const unsigned* __restrict__ keys, int shift, int nbins, unsigned mask, int* __restrict__ hist, int ntiles) {
  extern __shared__ int rs_lds[];
  radix_hist_body(n, keys, shift, nbins, mask, hist, ntiles, rs_lds);
}
// the same with the element count and the digit split on the DEVICE (SortP, written by the launch that sized the keys): the host fixes only the
// number of passes and the capacity; workgroups beyond the live tiles write the zeros the table scan expects
__device__ __forceinline__ void sortp_pass(const SortP& sp, int p, int& shift, unsigned& mask) {
  shift = sp.db * p;
  const int live = max(0, min(sp.db, sp.bits - sp.db * p));
  mask = (1u << live) - 1u;
}
__global__ __launch_bounds__(kRT) void k_radix_hist_dc(const SortP* __restrict__ spp, const unsigned* __restrict__ keys, int p, int* __restrict__ hist, int ntiles) {
  extern __shared__ int rs_lds[];
  const SortP sp = *spp;
  int shift; unsigned mask;
  sortp_pass(sp, p, shift, mask);
  radix_hist_body(sp.n, keys, shift, sp.nbins, mask, hist, ntiles, rs_lds);
}

__device__ __forceinline__ int wave_sum_i(int v) { for (int o = 32; o > 0; o >>= 1) v += __shfl_down(v, o); return v; }      // (lane 0 holds the sum)

// exclusive scan of a[lo0, hi0) in place starting from `start`, by one workgroup: a contiguous chunk per thread, wave scans of the chunk
// sums on the DPP path
__device__ __forceinline__ void wg_excl_scan(int* __restrict__ a, const int lo0, const int hi0, const int start) {
  __shared__ int s_wave[kRScanT / 64];
  const int total = hi0 - lo0;
  const int per = (total + kRScanT - 1) / kRScanT;
  const int lo = lo0 + min(total, (int)threadIdx.x * per), hi = min(hi0, lo + per);
  int sum = 0;
  for (int i = lo; i < hi; ++i) sum += a[i];
  const int incl = wave_incl_scan(sum);
  const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
  if (lane == 63) s_wave[w] = incl;
  __syncthreads();
  int basev = start;
  for (int k = 0; k < w; ++k) basev += s_wave[k];
  int run = basev + incl - sum;
  for (int i = lo; i < hi; ++i) { const int v = a[i]; a[i] = run; run += v; }
}
// small tables (a 100 k-point cloud with 10-bit digits: 25 k entries): ONE workgroup
__global__ __launch_bounds__(kRScanT) void k_radix_scan(int total, int* __restrict__ a) { wg_excl_scan(a, 0, total, 0); }
__global__ __launch_bounds__(kRScanT) void k_radix_scan_dc(const SortP* __restrict__ spp, int ntiles, int* __restrict__ a) { wg_excl_scan(a, 0, spp->nbins * ntiles, 0); }
// large tables (a 10 M-point cloud: 5 M entries — one workgroup walking them was the bottleneck of a pass): `chunk` entries per workgroup,
// two launches: the chunk sums, then every workgroup scans its chunk from the sum of the chunks before it (<= kRScanMaxGroups of them:
// each workgroup adds them up itself, no third launch)
constexpr int kRScanMaxGroups = 1024, kRScanOneGroupMax = 1 << 18;
__global__ __launch_bounds__(kRScanT) void k_radix_scan_sums(int total, const int* __restrict__ a, int chunk, int* __restrict__ part) {
  __shared__ int s_w[kRScanT / 64];
  const int lo = blockIdx.x * chunk, hi = min(total, lo + chunk);
  int sum = 0;
  for (int i = lo + (int)threadIdx.x; i < hi; i += kRScanT) sum += a[i];
  sum = wave_sum_i(sum);
  if ((threadIdx.x & 63) == 0) s_w[threadIdx.x >> 6] = sum;
  __syncthreads();
  if (threadIdx.x == 0) { int t = 0; for (int k = 0; k < kRScanT / 64; ++k) t += s_w[k]; part[blockIdx.x] = t; }
}
__global__ __launch_bounds__(kRScanT) void k_radix_scan_chunks(int total, int* __restrict__ a, int chunk, const int* __restrict__ part) {
  __shared__ int s_w[kRScanT / 64];
  __shared__ int s_start;
  int v = (int)threadIdx.x < (int)blockIdx.x ? part[threadIdx.x] : 0;      // (gridDim.x <= kRScanMaxGroups = the block size)
  v = wave_sum_i(v);
  if ((threadIdx.x & 63) == 0) s_w[threadIdx.x >> 6] = v;
  __syncthreads();
  if (threadIdx.x == 0) { int t = 0; for (int k = 0; k < kRScanT / 64; ++k) t += s_w[k]; s_start = t; }
  __syncthreads();
  const int lo = blockIdx.x * chunk;
  wg_excl_scan(a, lo, min(total, lo + chunk), s_start);
}

// Dynamic LDS: s_base[nbins] ints | s_cnt[waves][nbins] BYTES (a wave holds at most 64 keys of a digit).  1 024 threads = 16 waves: a tile is four
// rounds of 1 024 keys (with 256 threads it was sixteen rounds of four barriers each, 20 us per pass on a 50 k-key sort whose dozen tiles cannot
// fill the chip anyway).
constexpr int kRST = 1024, kRSRounds = kRTile / kRST, kRSWaves = kRST / 64;
static_assert(kRSRounds * kRST == kRTile, "a tile is a whole number of scatter rounds");
__host__ __device__ inline size_t radix_scatter_lds(int nbins) { return (size_t)nbins * sizeof(int) + (size_t)kRSWaves * nbins; }
__device__ __forceinline__ void radix_scatter_body(int n, const unsigned* __restrict__ keys_in, const int* __restrict__ vals_in, unsigned* __restrict__ keys_out,
                                                   int* __restrict__ vals_out, int shift, int digit_bits, unsigned mask, const int* __restrict__ hist, int ntiles, int* rs_lds) {
  const int nbins = 1 << digit_bits;
  int* s_base = rs_lds;
  unsigned char* s_cnt = reinterpret_cast<unsigned char*>(rs_lds + nbins);      // [wave][bin]
  unsigned* s_cnt_w = reinterpret_cast<unsigned*>(rs_lds + nbins);              // the same as words, for clearing
  for (int b = threadIdx.x; b < nbins; b += kRST) s_base[b] = hist[(size_t)b * ntiles + blockIdx.x];
  const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
  const unsigned long long lt = lane == 0 ? 0ull : (~0ull >> (64 - lane));
  const int base = blockIdx.x * kRTile;
  for (int r = 0; r < kRSRounds; ++r) {
    if (base + r * kRST >= n) break;                  // (uniform: the rest of the tile holds no keys)
    for (int b = threadIdx.x; b < kRSWaves * nbins / 4; b += kRST) s_cnt_w[b] = 0u;
    __syncthreads();
    const int i = base + r * kRST + threadIdx.x;
    const bool valid = i < n;
    const unsigned key = valid ? keys_in[i] : 0u;
    const int val = valid ? vals_in[i] : 0;
    const unsigned d = (key >> shift) & mask;
    unsigned long long same = __ballot(valid);          // lanes holding a key of MY digit
    for (int b = 0; b < digit_bits; ++b) {
      const bool bit = (d >> b) & 1u;
      const unsigned long long bal = __ballot(bit);
      same &= bit ? bal : ~bal;
    }
    const int rank = __popcll(same & lt), cnt = __popcll(same);
    if (valid && rank == 0) s_cnt[w * nbins + d] = (unsigned char)cnt;      // (<= 64)
    __syncthreads();
    if (valid) {
      int off = s_base[d] + rank;
      for (int k = 0; k < w; ++k) { const int c = s_cnt[k * nbins + d]; off += c; }
      keys_out[off] = key; vals_out[off] = val;
    }
    __syncthreads();
    for (int b = threadIdx.x; b < nbins; b += kRST) {
      int add = 0;
#pragma unroll
      for (int k = 0; k < kRSWaves; ++k) add += s_cnt[k * nbins + b];
      s_base[b] += add;
    }
    __syncthreads();
  }
}
__global__ __launch_bounds__(kRST) void k_radix_scatter(int n, const unsigned* __restrict__ keys_in, const int* __restrict__ vals_in, unsigned* __restrict__ keys_out,
                                                       int* __restrict__ vals_out, int shift, int digit_bits, unsigned mask, const int* __restrict__ hist, int ntiles) {
  extern __shared__ int rs_lds[];
  radix_scatter_body(n, keys_in, vals_in, keys_out, vals_out, shift, digit_bits, mask, hist, ntiles, rs_lds);
}
__global__ __launch_bounds__(kRST) void k_radix_scatter_dc(const SortP* __restrict__ spp, const unsigned* __restrict__ keys_in, const int* __restrict__ vals_in,
                                                          unsigned* __restrict__ keys_out, int* __restrict__ vals_out, int p, const int* __restrict__ hist, int ntiles) {
  extern __shared__ int rs_lds[];
  const SortP sp = *spp;
  if ((long long)blockIdx.x * kRTile >= (long long)sp.n) return;      // (a tile without keys: uniform per workgroup)
  int shift; unsigned mask;
  sortp_pass(sp, p, shift, mask);
  radix_scatter_body(sp.n, keys_in, vals_in, keys_out, vals_out, shift, sp.db, mask, hist, ntiles, rs_lds);
}

int device_sort_pairs_u32(lvf_ctx* ctx, const unsigned* keys_in, unsigned* keys_out, const int* vals_in, int* vals_out, int n, int key_bits) {
  if (n <= 0) return LVF_OK;
  hipStream_t s = ctx->stream;
  const int bits = std::min(32, std::max(1, key_bits));
  const int passes = (bits + kRMaxDigitBits - 1) / kRMaxDigitBits;      // as few passes as 11-bit digits allow ...
  const int db = (bits + passes - 1) / passes;                          // ... each as narrow as that number of passes permits
  const int nbins = 1 << db;
  const int ntiles = (n + kRTile - 1) / kRTile;
  DevBuf<int> hist, part; DevBuf<unsigned> tkeys; DevBuf<int> tvals;
  LVF_TRY(hist.alloc((size_t)nbins * ntiles));
  const int total = nbins * ntiles;
  const int groups = total <= kRScanOneGroupMax ? 1 : std::min(kRScanMaxGroups, (total + kRScanOneGroupMax / 4 - 1) / (kRScanOneGroupMax / 4));
  const int chunk = (total + groups - 1) / groups;
  if (groups > 1) LVF_TRY(part.alloc(groups));
  if (passes > 1) { LVF_TRY(tkeys.alloc(n)); LVF_TRY(tvals.alloc(n)); }
  const unsigned* ki = keys_in; const int* vi = vals_in;
  for (int p = 0; p < passes; ++p) {
    // the last pass must land in the caller's arrays: with an odd number of passes the first one writes there, otherwise the scratch pair
    const bool to_out = ((passes - 1 - p) % 2) == 0;
    unsigned* ko = to_out ? keys_out : tkeys.p; int* vo = to_out ? vals_out : tvals.p;
    // the digit's live bits: all db of them, except in the last pass when db * passes > bits (13-bit keys: digits of 7 and 6 bits)
    const int live = std::min(db, bits - db * p);
    const unsigned mask = (1u << live) - 1u;
    hipLaunchKernelGGL(k_radix_hist, dim3(ntiles), dim3(kRT), (size_t)nbins * sizeof(int), s, n, ki, db * p, nbins, mask, hist.p, ntiles);
    if (groups == 1) hipLaunchKernelGGL(k_radix_scan, dim3(1), dim3(kRScanT), 0, s, total, hist.p);
    else {
      hipLaunchKernelGGL(k_radix_scan_sums, dim3(groups), dim3(kRScanT), 0, s, total, hist.p, chunk, part.p);
      hipLaunchKernelGGL(k_radix_scan_chunks, dim3(groups), dim3(kRScanT), 0, s, total, hist.p, chunk, part.p);
    }
    hipLaunchKernelGGL(k_radix_scatter, dim3(ntiles), dim3(kRST), radix_scatter_lds(nbins), s, n, ki, vi, ko, vo, db * p, db, mask, hist.p, ntiles);
    ki = ko; vi = vo;
  }
  LVF_HIP(hipGetLastError());
  // (the scratch buffers go back to the context's pool on return: the pool hands them out again in stream order, nothing to wait for)
  return LVF_OK;
}

// The sort with the count and the key width on the device.  `passes` digit passes are enqueued whatever the keys turn out to need (the launch
// that wrote *sp raises its error flag when they need more); `cap` = the arrays' capacity.  Several independent sorts (each on its own stream)
// are enqueued step by step, one launch of each in turn, so that none of them waits for the host to finish enqueueing the other.
int device_sort_pairs_u32_dc_multi(lvf_ctx* ctx, int n_jobs, SortJobDc* jobs, int cap, int passes) {
  (void)ctx;
  if (cap <= 0) return LVF_OK;
  const int nbins_max = 1 << kRMaxDigitBits;
  const int ntiles = (cap + kRTile - 1) / kRTile;
  if ((long long)nbins_max * ntiles > kRScanOneGroupMax) { set_error("device_sort_pairs_u32_dc: %d keys are beyond the one-workgroup table scan", cap); return LVF_ERR_INVALID; }
  for (int k = 0; k < n_jobs; ++k) {
    SortScratch& K = *jobs[k].keep;
    LVF_TRY(K.hist.alloc((size_t)nbins_max * ntiles));
    if (passes > 1) { LVF_TRY(K.tkeys.alloc(cap)); LVF_TRY(K.tvals.alloc(cap)); }
    jobs[k].ki = jobs[k].keys_in; jobs[k].vi = jobs[k].vals_in;
  }
  for (int p = 0; p < passes; ++p) {
    const bool to_out = ((passes - 1 - p) % 2) == 0;
    for (int k = 0; k < n_jobs; ++k) {
      SortJobDc& J = jobs[k];
      J.ko = to_out ? J.keys_out : J.keep->tkeys.p; J.vo = to_out ? J.vals_out : J.keep->tvals.p;
      hipLaunchKernelGGL(k_radix_hist_dc, dim3(ntiles), dim3(kRT), (size_t)nbins_max * sizeof(int), J.q, J.sp, J.ki, p, J.keep->hist.p, ntiles);
    }
    for (int k = 0; k < n_jobs; ++k) hipLaunchKernelGGL(k_radix_scan_dc, dim3(1), dim3(kRScanT), 0, jobs[k].q, jobs[k].sp, ntiles, jobs[k].keep->hist.p);
    for (int k = 0; k < n_jobs; ++k) {
      SortJobDc& J = jobs[k];
      hipLaunchKernelGGL(k_radix_scatter_dc, dim3(ntiles), dim3(kRST), radix_scatter_lds(nbins_max), J.q, J.sp, J.ki, J.vi, J.ko, J.vo, p, J.keep->hist.p, ntiles);
      J.ki = J.ko; J.vi = J.vo;
    }
  }
  LVF_HIP(hipGetLastError());
  return LVF_OK;
}

// ---- exclusive scan / order-preserving compaction in ONE launch, element count on the device ---------------------------------------------
// A tile of 2 048 flags per workgroup.  Tiles are handed out by a ticket (so a tile only ever waits for tiles that already run), every tile
// publishes its sum as {epoch, sum} — a relaxed agent-scope atomic store of one 64-bit word: the word IS the message, nothing else has to be
// made visible, so nobody fences — and the first wave of a tile reads the sums of ALL the tiles before it, 64 per load (a cloud of 115 k points
// is 57 tiles: one load).  The epoch (a per-context launch counter) makes the status array self-cleaning; the tile holding the last ticket puts
// the ticket counter back to 0.  Replaces three scan launches, the compaction launch and — where the count is only needed by later launches —
// the read-back between them.
constexpr int kS1T = 256, kS1E = 8, kS1Tile = kS1T * kS1E;
template <bool COMPACT>
__global__ __launch_bounds__(kS1T) void k_scan1(int cap, const int* __restrict__ n_dev, const int* __restrict__ in, int* __restrict__ pos, int* __restrict__ total_out,
                                                unsigned long long* status, int* ticket, unsigned epoch, int ntiles, const float4* __restrict__ pts,
                                                float4* __restrict__ out, unsigned long long* __restrict__ err) {
  __shared__ int s_tile, s_prefix, s_wave[kS1T / 64];
  if (threadIdx.x == 0) s_tile = atomicAdd(ticket, 1);
  __syncthreads();
  const int tile = s_tile;
  const int n = n_dev ? min(cap, *n_dev) : cap;
  const int base = tile * kS1Tile + (int)threadIdx.x * kS1E;
  int v[kS1E];
  int sum = 0;
#pragma unroll
  for (int e = 0; e < kS1E; ++e) { v[e] = (base + e < n) ? in[base + e] : 0; sum += v[e]; }
  const int incl = wave_incl_scan(sum);
  const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
  if (lane == 63) s_wave[w] = incl;
  __syncthreads();
  int wbase = 0, tot = 0;
#pragma unroll
  for (int k = 0; k < kS1T / 64; ++k) { const int q = s_wave[k]; if (k < w) wbase += q; tot += q; }
  if (threadIdx.x == 0) {
    __hip_atomic_store(status + tile, ((unsigned long long)epoch << 32) | (unsigned long long)(unsigned)tot, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    if (tile == ntiles - 1) __hip_atomic_store(ticket, 0, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);      // (every ticket of this launch has been taken)
  }
  if (w == 0) {
    int acc = 0;
    for (int t = lane; t < tile; t += 64) {
      unsigned long long st;
      unsigned polls = 0;
      do {
        st = __hip_atomic_load(status + t, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        // (seconds without the predecessor's word: the status block was corrupted, or two host threads drove ONE context — tickets are per
        // context.  The launch is NOT trapped (that would take the queue, i.e. the process, down with it): the sticky error word is raised, this
        // tile goes on with what it has, and the host's next count read-back of this context fails with LVF_ERR_STATE — read_back())
        if (++polls > (1u << 23)) { atomicExch(err, ((unsigned long long)epoch << 32) | (unsigned long long)(unsigned)(t + 1)); break; }
      } while ((unsigned)(st >> 32) != epoch);
      acc += (int)(unsigned)(st & 0xffffffffull);
    }
    acc = wave_sum_i(acc);
    if (lane == 0) s_prefix = acc;
  }
  __syncthreads();
  const int prefix = s_prefix;
  int run = prefix + wbase + incl - sum;
  if (COMPACT) {
    // the thread's records are requested together (index clamped), then stored where flagged: as `if (flag) out[..] = pts[i]` every load sat
    // in its own branch and was waited for before its store — eight dependent round trips per thread
    if (base < n && sum != 0) {
      float4 pv[kS1E];
#pragma unroll
      for (int e = 0; e < kS1E; ++e) pv[e] = pts[min(base + e, n - 1)];
#pragma unroll
      for (int e = 0; e < kS1E; ++e) {
        if (pos && base + e < n) pos[base + e] = run;
        if (v[e]) out[run] = pv[e];
        run += v[e];
      }
    } else if (pos) {
#pragma unroll
      for (int e = 0; e < kS1E; ++e) if (base + e < n) pos[base + e] = run;      // (no flag set: every position of the thread has the same prefix)
    }
  } else {
#pragma unroll
    for (int e = 0; e < kS1E; ++e) {
      const int i = base + e;
      if (i < n && pos) pos[i] = run;
      run += v[e];
    }
  }
  if (tile == ntiles - 1 && threadIdx.x == 0) {
    if (total_out) *total_out = prefix + tot;
    if (pos) pos[n] = prefix + tot;
  }
}
// in[0 .. n) -> pos[0 .. n] (optional; pos[n] = the total), *total_out (optional), and — with pts / out — out[pos[i]] = pts[i] where in[i] != 0.
// n = *n_dev clipped to cap, or cap when n_dev is null.
int device_scan1_on(lvf_ctx* ctx, hipStream_t s, int lane, const int* in, int cap, const int* n_dev, int* pos, int* total_out, const float4* pts, float4* out) {
  const int ntiles = std::max(1, (cap + kS1Tile - 1) / kS1Tile);
  if (ntiles > ctx->scan_tiles) {
    LVF_HIP(hipStreamSynchronize(ctx->stream));
    if (ctx->stream2) LVF_HIP(hipStreamSynchronize(ctx->stream2));
    if (ctx->scan_status) (void)hipFree(ctx->scan_status);
    ctx->scan_status = nullptr; ctx->scan_tiles = 0; ctx->scan_err = nullptr;
    const int want = std::max(1024, ntiles + ntiles / 2);
    const size_t words = (size_t)2 * want + 3;            // [lane][tile] status words, one ticket word per lane, then the sticky error word
    LVF_HIP(hipMalloc(reinterpret_cast<void**>(&ctx->scan_status), words * sizeof(unsigned long long)));
    // (on the context's stream and waited for: a plain hipMemset goes to the null stream, which a non-blocking stream does not order against —
    // the clear could land in the middle of the launch below, hand a ticket out twice and leave a tile waiting for ever)
    LVF_HIP(hipMemsetAsync(ctx->scan_status, 0, words * sizeof(unsigned long long), ctx->stream));
    LVF_HIP(hipStreamSynchronize(ctx->stream));
    ctx->scan_tiles = want; ctx->scan_epoch = 0;
  }
  unsigned long long* status = ctx->scan_status + (size_t)lane * ctx->scan_tiles;
  int* ticket = reinterpret_cast<int*>(ctx->scan_status + (size_t)2 * ctx->scan_tiles + lane);
  unsigned epoch = ++ctx->scan_epoch;
  if (epoch == 0) {          // the counter wrapped: every stale word could match again
    LVF_HIP(hipStreamSynchronize(ctx->stream));
    if (ctx->stream2) LVF_HIP(hipStreamSynchronize(ctx->stream2));
    LVF_HIP(hipMemsetAsync(ctx->scan_status, 0, ((size_t)2 * ctx->scan_tiles + 3) * sizeof(unsigned long long), ctx->stream));
    LVF_HIP(hipStreamSynchronize(ctx->stream));
    epoch = ctx->scan_epoch = 1;
  }
  unsigned long long* err = ctx->scan_status + (size_t)2 * ctx->scan_tiles + 2;
  ctx->scan_err = err;                  // read_back() looks at it on the next count read-back of this context
  if (pts && out) hipLaunchKernelGGL(k_scan1<true>, dim3(ntiles), dim3(kS1T), 0, s, cap, n_dev, in, pos, total_out, status, ticket, epoch, ntiles, pts, out, err);
  else hipLaunchKernelGGL(k_scan1<false>, dim3(ntiles), dim3(kS1T), 0, s, cap, n_dev, in, pos, total_out, status, ticket, epoch, ntiles, pts, out, err);
  LVF_HIP(hipGetLastError());
  return LVF_OK;
}
int device_scan1(lvf_ctx* ctx, const int* in, int cap, const int* n_dev, int* pos, int* total_out, const float4* pts, float4* out) {
  return device_scan1_on(ctx, ctx->stream, 0, in, cap, n_dev, pos, total_out, pts, out);
}
int side_stream(lvf_ctx* ctx, hipStream_t* out) {
  if (!ctx->stream2) {
    LVF_HIP(hipStreamCreateWithFlags(&ctx->stream2, hipStreamNonBlocking));
    LVF_HIP(hipEventCreateWithFlags(&ctx->ev_fork, hipEventDisableTiming));
    LVF_HIP(hipEventCreateWithFlags(&ctx->ev_join, hipEventDisableTiming));
  }
  *out = ctx->stream2;
  return LVF_OK;
}

}  // namespace lvf

extern "C" {
// Test hook (tests/test_gpu_cloud.py): the stable pair sort under VoxelGrid on host arrays.  keys_out / vals_out [n]; key_bits as in
// device_sort_pairs_u32 (bits above it must not influence the order).  Not part of the reference surface.
int lvf_debug_sort_pairs_u32(lvf_ctx* ctx, const uint32_t* keys, const int32_t* vals, int n, int key_bits, uint32_t* keys_out, int32_t* vals_out) {
  LVF_REQUIRE(ctx && n >= 0 && (n == 0 || (keys && vals && keys_out && vals_out)) && key_bits >= 1 && key_bits <= 32, "lvf_debug_sort_pairs_u32: bad arguments");
  if (n == 0) return LVF_OK;
  LVF_TRY(lvf::enter(ctx));
  hipStream_t s = ctx->stream;
  lvf::DevBuf<unsigned> ki, ko; lvf::DevBuf<int> vi, vo;
  LVF_TRY(ki.alloc(n)); LVF_TRY(ko.alloc(n)); LVF_TRY(vi.alloc(n)); LVF_TRY(vo.alloc(n));
  LVF_HIP(hipMemcpyAsync(ki.p, keys, (size_t)n * 4, hipMemcpyHostToDevice, s));
  LVF_HIP(hipMemcpyAsync(vi.p, vals, (size_t)n * 4, hipMemcpyHostToDevice, s));
  LVF_HIP(hipStreamSynchronize(s));          // (pageable sources: the copies above must not outlive this call's arguments)
  LVF_TRY(lvf::device_sort_pairs_u32(ctx, ki.p, ko.p, vi.p, vo.p, n, key_bits));
  LVF_HIP(hipMemcpyAsync(keys_out, ko.p, (size_t)n * 4, hipMemcpyDeviceToHost, s));
  LVF_HIP(hipMemcpyAsync(vals_out, vo.p, (size_t)n * 4, hipMemcpyDeviceToHost, s));
  LVF_HIP(hipStreamSynchronize(s));
  return LVF_OK;
}
}

