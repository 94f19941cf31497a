// libprosstt_amd_host.so: host-side helpers of the drop-in path (include/prosstt_amd_host.h).  No HIP in here.
//
// int32 -> int64 on a pool of worker threads.  The result of a 50 000 x 20 000 call is 8 GB: written once, never read
// back by this code, so the AVX2 form stores past the caches (non-temporal: no read-for-ownership of 8 GB), and the
// pool is kept between calls (a chunk of 256 MB is widened in a few milliseconds; starting threads per chunk would show).
#include <atomic>
#include <condition_variable>
#include <cstdint>
#include <cstring>
#include <mutex>
#include <new>
#include <thread>
#include <vector>

#include <immintrin.h>
#include <pthread.h>

#include "../../../include/prosstt_amd_host.h"

#define PH_EXPORT extern "C" __attribute__((visibility("default")))

namespace {

void widen_plain(const int32_t* src, int64_t* dst, uint64_t n)
{
    for (uint64_t i = 0; i < n; ++i) dst[i] = (int64_t)src[i];
}

void widen16_plain(const uint16_t* src, int64_t* dst, uint64_t n)
{
    for (uint64_t i = 0; i < n; ++i) dst[i] = (int64_t)src[i];
}

void widen16to32_plain(const uint16_t* src, int32_t* dst, uint64_t n)
{
    for (uint64_t i = 0; i < n; ++i) dst[i] = (int32_t)src[i];
}

void widen8_plain(const uint8_t* src, int64_t* dst, uint64_t n)
{
    for (uint64_t i = 0; i < n; ++i) dst[i] = (int64_t)src[i];
}

void widen8to32_plain(const uint8_t* src, int32_t* dst, uint64_t n)
{
    for (uint64_t i = 0; i < n; ++i) dst[i] = (int32_t)src[i];
}

__attribute__((target("avx2"))) void widen8_avx2(const uint8_t* src, int64_t* dst, uint64_t n)
{
    uint64_t i = 0;
    while (i < n && (reinterpret_cast<uintptr_t>(dst + i) & 31u) != 0u) { dst[i] = (int64_t)src[i]; ++i; }
    for (; i + 16 <= n; i += 16) {
        const __m128i a = _mm_loadu_si128(reinterpret_cast<const __m128i*>(src + i));           // 16 counts
        _mm256_stream_si256(reinterpret_cast<__m256i*>(dst + i), _mm256_cvtepu8_epi64(a));
        _mm256_stream_si256(reinterpret_cast<__m256i*>(dst + i + 4), _mm256_cvtepu8_epi64(_mm_srli_si128(a, 4)));
        _mm256_stream_si256(reinterpret_cast<__m256i*>(dst + i + 8), _mm256_cvtepu8_epi64(_mm_srli_si128(a, 8)));
        _mm256_stream_si256(reinterpret_cast<__m256i*>(dst + i + 12), _mm256_cvtepu8_epi64(_mm_srli_si128(a, 12)));
    }
    for (; i < n; ++i) dst[i] = (int64_t)src[i];
    _mm_sfence();
}

__attribute__((target("avx2"))) void widen8to32_avx2(const uint8_t* src, int32_t* dst, uint64_t n)
{
    uint64_t i = 0;
    while (i < n && (reinterpret_cast<uintptr_t>(dst + i) & 31u) != 0u) { dst[i] = (int32_t)src[i]; ++i; }
    for (; i + 16 <= n; i += 16) {
        const __m128i a = _mm_loadu_si128(reinterpret_cast<const __m128i*>(src + i));
        _mm256_stream_si256(reinterpret_cast<__m256i*>(dst + i), _mm256_cvtepu8_epi32(a));
        _mm256_stream_si256(reinterpret_cast<__m256i*>(dst + i + 8), _mm256_cvtepu8_epi32(_mm_srli_si128(a, 8)));
    }
    for (; i < n; ++i) dst[i] = (int32_t)src[i];
    _mm_sfence();
}

// dst[pos[i]] = val[i] for the entries [lo, lo + n) of a list (kinds 5 and 6: an int64 / int32 destination); `src` is the
// list of values, `aux` the positions
void scatter(const int32_t* val, const int64_t* pos, void* dst, bool wide, uint64_t lo, uint64_t n)
{
    if (wide) for (uint64_t i = lo; i < lo + n; ++i) static_cast<int64_t*>(dst)[pos[i]] = (int64_t)val[i];
    else for (uint64_t i = lo; i < lo + n; ++i) static_cast<int32_t*>(dst)[pos[i]] = val[i];
}

__attribute__((target("avx2"))) void widen16_avx2(const uint16_t* src, int64_t* dst, uint64_t n)
{
    uint64_t i = 0;
    while (i < n && (reinterpret_cast<uintptr_t>(dst + i) & 31u) != 0u) { dst[i] = (int64_t)src[i]; ++i; }
    for (; i + 16 <= n; i += 16) {
        const __m128i a = _mm_loadu_si128(reinterpret_cast<const __m128i*>(src + i));           // 8 counts
        const __m128i b = _mm_loadu_si128(reinterpret_cast<const __m128i*>(src + i + 8));
        _mm256_stream_si256(reinterpret_cast<__m256i*>(dst + i), _mm256_cvtepu16_epi64(a));
        _mm256_stream_si256(reinterpret_cast<__m256i*>(dst + i + 4), _mm256_cvtepu16_epi64(_mm_srli_si128(a, 8)));
        _mm256_stream_si256(reinterpret_cast<__m256i*>(dst + i + 8), _mm256_cvtepu16_epi64(b));
        _mm256_stream_si256(reinterpret_cast<__m256i*>(dst + i + 12), _mm256_cvtepu16_epi64(_mm_srli_si128(b, 8)));
    }
    for (; i < n; ++i) dst[i] = (int64_t)src[i];
    _mm_sfence();
}

__attribute__((target("avx2"))) void widen16to32_avx2(const uint16_t* src, int32_t* dst, uint64_t n)
{
    uint64_t i = 0;
    while (i < n && (reinterpret_cast<uintptr_t>(dst + i) & 31u) != 0u) { dst[i] = (int32_t)src[i]; ++i; }
    for (; i + 16 <= n; i += 16) {
        const __m128i a = _mm_loadu_si128(reinterpret_cast<const __m128i*>(src + i));
        const __m128i b = _mm_loadu_si128(reinterpret_cast<const __m128i*>(src + i + 8));
        _mm256_stream_si256(reinterpret_cast<__m256i*>(dst + i), _mm256_cvtepu16_epi32(a));
        _mm256_stream_si256(reinterpret_cast<__m256i*>(dst + i + 8), _mm256_cvtepu16_epi32(b));
    }
    for (; i < n; ++i) dst[i] = (int32_t)src[i];
    _mm_sfence();
}

__attribute__((target("avx2"))) void widen_avx2(const int32_t* src, int64_t* dst, uint64_t n)
{
    uint64_t i = 0;
    // up to a 32-byte boundary of the destination
    while (i < n && (reinterpret_cast<uintptr_t>(dst + i) & 31u) != 0u) { dst[i] = (int64_t)src[i]; ++i; }
    for (; i + 16 <= n; i += 16) {
        const __m128i a = _mm_loadu_si128(reinterpret_cast<const __m128i*>(src + i));
        const __m128i b = _mm_loadu_si128(reinterpret_cast<const __m128i*>(src + i + 4));
        const __m128i c = _mm_loadu_si128(reinterpret_cast<const __m128i*>(src + i + 8));
        const __m128i d = _mm_loadu_si128(reinterpret_cast<const __m128i*>(src + i + 12));
        _mm256_stream_si256(reinterpret_cast<__m256i*>(dst + i), _mm256_cvtepi32_epi64(a));
        _mm256_stream_si256(reinterpret_cast<__m256i*>(dst + i + 4), _mm256_cvtepi32_epi64(b));
        _mm256_stream_si256(reinterpret_cast<__m256i*>(dst + i + 8), _mm256_cvtepi32_epi64(c));
        _mm256_stream_si256(reinterpret_cast<__m256i*>(dst + i + 12), _mm256_cvtepi32_epi64(d));
    }
    for (; i < n; ++i) dst[i] = (int64_t)src[i];
    _mm_sfence();
}

bool has_avx2()
{
    static const bool yes = __builtin_cpu_supports("avx2");
    return yes;
}

// A pool of workers that all run the same job on their own slice; the caller is worker 0.
class Pool {
public:
    ~Pool()
    {
        {
            std::lock_guard<std::mutex> g(m_);
            quit_ = true;
        }
        wake_.notify_all();
        for (auto& t : threads_) t.join();
    }

    // kind: 0 int32 -> int64, 1 uint16 -> int64, 2 uint16 -> int32, 3 uint8 -> int64, 4 uint8 -> int32,
    //       5 / 6 scatter of int32 values to the positions `aux` of an int64 / int32 destination
    void run(const void* src, void* dst, uint64_t count, int workers, int kind, const void* aux = nullptr)
    {
        std::lock_guard<std::mutex> serial(call_);
        grow(workers - 1);
        {
            std::lock_guard<std::mutex> g(m_);
            src_ = src; dst_ = dst; count_ = count; workers_ = workers; kind_ = kind; aux_ = aux;
            next_.store(0, std::memory_order_relaxed);
            pending_ = workers - 1;
            ++epoch_;
        }
        wake_.notify_all();
        slice(0);
        std::unique_lock<std::mutex> g(m_);
        done_.wait(g, [&] { return pending_ == 0; });
    }

private:
    void grow(int n)
    {
        while ((int)threads_.size() < n) {
            const int id = (int)threads_.size() + 1;
            uint64_t seen;
            {
                std::lock_guard<std::mutex> g(m_);
                seen = epoch_;
            }
            threads_.emplace_back([this, id, seen] { loop(id, seen); });
        }
    }

    void slice(int)
    {
        // Pieces of kPiece counts, handed out by one counter: the workers are not pinned, and one that shares its core or
        // sits a socket away from the memory would otherwise decide how long the whole job takes.  (Pieces start on
        // multiples of 2^17 counts, so every piece but a misaligned first starts on a cache line of dst.)
        for (;;) {
            const uint64_t lo = next_.fetch_add(kPiece, std::memory_order_relaxed);
            if (lo >= count_) return;
            const uint64_t n = count_ - lo < kPiece ? count_ - lo : kPiece;
            convert(kind_, src_, dst_, lo, n, aux_);
        }
    }

public:
    static void convert(int kind, const void* src, void* dst, uint64_t lo, uint64_t n, const void* aux = nullptr)
    {
        if (kind >= 5)
            scatter(static_cast<const int32_t*>(src), static_cast<const int64_t*>(aux), dst, kind == 5, lo, n);
        else if (kind == 3)
            (has_avx2() ? widen8_avx2 : widen8_plain)(static_cast<const uint8_t*>(src) + lo, static_cast<int64_t*>(dst) + lo, n);
        else if (kind == 4)
            (has_avx2() ? widen8to32_avx2 : widen8to32_plain)(static_cast<const uint8_t*>(src) + lo, static_cast<int32_t*>(dst) + lo, n);
        else if (kind == 0)
            (has_avx2() ? widen_avx2 : widen_plain)(static_cast<const int32_t*>(src) + lo, static_cast<int64_t*>(dst) + lo, n);
        else if (kind == 1)
            (has_avx2() ? widen16_avx2 : widen16_plain)(static_cast<const uint16_t*>(src) + lo, static_cast<int64_t*>(dst) + lo, n);
        else
            (has_avx2() ? widen16to32_avx2 : widen16to32_plain)(static_cast<const uint16_t*>(src) + lo, static_cast<int32_t*>(dst) + lo, n);
    }

private:

    void loop(int id, uint64_t seen)
    {
        for (;;) {
            std::unique_lock<std::mutex> g(m_);
            wake_.wait(g, [&] { return quit_ || epoch_ != seen; });
            if (quit_) return;
            seen = epoch_;
            const bool mine = id < workers_;
            g.unlock();
            if (mine) {
                slice(id);
                g.lock();
                if (--pending_ == 0) done_.notify_one();
            }
        }
    }

    static constexpr uint64_t kPiece = 1u << 17;
    std::atomic<uint64_t> next_{0};
    std::mutex call_, m_;
    std::condition_variable wake_, done_;
    std::vector<std::thread> threads_;
    const void* src_ = nullptr;
    const void* aux_ = nullptr;
    void* dst_ = nullptr;
    uint64_t count_ = 0, epoch_ = 0;
    int workers_ = 1, pending_ = 0, kind_ = 0;
    bool quit_ = false;
};

// The pool is never destroyed (worker threads must not be joined from a library destructor at exit), and a forked child
// starts without one: the threads of the parent do not exist there.
Pool* g_pool = nullptr;
std::mutex g_pool_m;

void forget_pool_in_child()
{
    g_pool = nullptr;
    new (&g_pool_m) std::mutex();
}

Pool& pool()
{
    std::lock_guard<std::mutex> g(g_pool_m);
    if (!g_pool) {
        static const int registered = pthread_atfork(nullptr, nullptr, forget_pool_in_child);
        (void)registered;
        g_pool = new Pool();
    }
    return *g_pool;
}

}  // namespace

namespace {

int convert_on_pool(const void* src, void* dst, uint64_t count, int32_t threads, int kind, const void* aux = nullptr)
{
    if (count == 0) return 0;
    if (!src || !dst || (kind >= 5 && !aux)) return -1;
    int workers = threads < 1 ? 1 : (threads > 64 ? 64 : threads);
    // nothing to share out below a few pages per worker
    const uint64_t per = 1u << 14;
    if ((uint64_t)workers > (count + per - 1) / per) workers = (int)((count + per - 1) / per);
    if (workers <= 1) {
        Pool::convert(kind, src, dst, 0, count, aux);
        return 0;
    }
    try {
        pool().run(src, dst, count, workers, kind, aux);
    } catch (...) {                      // (thread creation failed: the caller's thread does all of it)
        Pool::convert(kind, src, dst, 0, count, aux);
    }
    return 0;
}

}  // namespace

PH_EXPORT int prosstt_amd_host_widen_i32_i64(const int32_t* src, int64_t* dst, uint64_t count, int32_t threads)
{
    return convert_on_pool(src, dst, count, threads, 0);
}

PH_EXPORT int prosstt_amd_host_widen_u16_i64(const uint16_t* src, int64_t* dst, uint64_t count, int32_t threads)
{
    return convert_on_pool(src, dst, count, threads, 1);
}

PH_EXPORT int prosstt_amd_host_widen_u16_i32(const uint16_t* src, int32_t* dst, uint64_t count, int32_t threads)
{
    return convert_on_pool(src, dst, count, threads, 2);
}

PH_EXPORT int prosstt_amd_host_widen_u8_i64(const uint8_t* src, int64_t* dst, uint64_t count, int32_t threads)
{
    return convert_on_pool(src, dst, count, threads, 3);
}

PH_EXPORT int prosstt_amd_host_widen_u8_i32(const uint8_t* src, int32_t* dst, uint64_t count, int32_t threads)
{
    return convert_on_pool(src, dst, count, threads, 4);
}

PH_EXPORT int prosstt_amd_host_scatter_i32(void* dst, int32_t dst_itemsize, const int64_t* positions, const int32_t* values,
                                           uint64_t count, int32_t threads)
{
    if (dst_itemsize != 8 && dst_itemsize != 4) return -1;
    return convert_on_pool(values, dst, count, threads, dst_itemsize == 8 ? 5 : 6, positions);
}

PH_EXPORT int prosstt_amd_host_has_avx2(void)
{
    return has_avx2() ? 1 : 0;
}
