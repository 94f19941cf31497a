// Host side of the lineage stage: the variates of a whole batch of candidate expression programs, drawn from
// numpy's GLOBAL legacy stream in the order the reference draws them, in one call.
//
// The reference's simulate_lineage (simulation.py:264-282) redraws a branch's K random walks until they are
// accepted; each walk (simulation.diffusion, simulation.py:104-113) takes from numpy's RandomState, in this order,
//     random.uniform(0, 1.5)   -> start        one 53-bit double
//     random.normal(0, 0.2)    -> first velocity    legacy polar Gaussian (with its cached second value)
//     random.uniform(0, 1)     -> momentum eta
//     random.normal(0, 2/T, T-1) -> velocity noise
// At C5 (256 branches, 3 770 attempts of 25 walks) that is 380 000 numpy calls from Python: 2 s of a stage whose
// device side takes 50 ms.  Here the caller hands over the generator's state (np.random.get_state()), one call
// produces the variates of B attempts and the state BEHIND EACH attempt, and the caller puts numpy's generator
// (np.random.set_state) behind the attempt it accepts: same numbers, same stream position as the one-at-a-time loop.
//
// numpy's legacy stream is frozen by NEP 19: MT19937 (Matsumoto & Nishimura 1998) with the reference tempering,
// doubles from two outputs as (a >> 5) * 2^26 + (b >> 6) over 2^53, normals by the Marsaglia polar method that keeps
// its second value for the next call.  tests/test_numpy_stream.py holds this file against numpy itself.
#pragma once
#include <cmath>
#include <cstdint>
#include <cstring>

namespace npstream {

constexpr int kWords = 624, kShift = 397;

struct Generator {
    uint32_t word[kWords];
    int32_t next;          // index of the next word to temper; kWords = the block is used up
    int32_t has_spare;     // the polar method's second normal is waiting
    double spare;

    void twist()
    {
        auto mix = [](uint32_t hi, uint32_t lo) {
            const uint32_t y = (hi & 0x80000000u) | (lo & 0x7fffffffu);
            return (y >> 1) ^ ((lo & 1u) ? 0x9908b0dfu : 0u);
        };
        for (int i = 0; i < kWords; ++i)
            word[i] = word[(i + kShift) % kWords] ^ mix(word[i], word[(i + 1) % kWords]);
        next = 0;
    }
    uint32_t bits32()
    {
        if (next >= kWords) twist();
        uint32_t y = word[next++];
        y ^= y >> 11;
        y ^= (y << 7) & 0x9d2c5680u;
        y ^= (y << 15) & 0xefc60000u;
        return y ^ (y >> 18);
    }
    double uniform()                       // random_sample(): [0, 1) with 53 bits
    {
        const uint32_t a = bits32() >> 5, b = bits32() >> 6;
        return ((double)a * 67108864.0 + (double)b) / 9007199254740992.0;
    }
    double normal()                        // standard_normal(): polar method, one value kept for the next call
    {
        if (has_spare) {
            has_spare = 0;
            const double v = spare;
            spare = 0.0;
            return v;
        }
        double x, y, r2;
        do {
            x = 2.0 * uniform() - 1.0;
            y = 2.0 * uniform() - 1.0;
            r2 = x * x + y * y;
        } while (r2 >= 1.0 || r2 == 0.0);
        const double f = std::sqrt(-2.0 * std::log(r2) / r2);
        spare = f * x;
        has_spare = 1;
        return f * y;
    }
};

// Variates of `attempts` consecutive sim_expr_branch(T, K) calls.  Layouts: start, vel0, eta [attempts][K];
// noise [attempts][K][T-1] (already scaled by 2/T).  after_* receive the generator's state behind every attempt.
inline void draw_programs(Generator& g, int32_t attempts, int32_t T, int32_t K, double* start, double* vel0, double* eta,
                          double* noise, uint32_t* after_words, int32_t* after_next, int32_t* after_has_spare,
                          double* after_spare)
{
    const double sigma = 2.0 / (double)T;
    const int64_t steps = T > 1 ? T - 1 : 0;
    for (int32_t a = 0; a < attempts; ++a) {
        for (int32_t k = 0; k < K; ++k) {
            const int64_t at = (int64_t)a * K + k;
            start[at] = g.uniform() * 1.5 + 0.0;
            vel0[at] = g.normal() * 0.2 + 0.0;
            eta[at] = g.uniform() * 1.0 + 0.0;
            double* e = noise + at * steps;
            for (int64_t t = 0; t < steps; ++t) e[t] = g.normal() * sigma + 0.0;
        }
        std::memcpy(after_words + (int64_t)a * kWords, g.word, sizeof(g.word));
        after_next[a] = g.next;
        after_has_spare[a] = g.has_spare;
        after_spare[a] = g.spare;
    }
}

}  // namespace npstream
