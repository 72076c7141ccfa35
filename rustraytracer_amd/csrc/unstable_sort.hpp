// Tie-order-exact stand-in for Rust's `slice::sort_unstable_by` (libcore's
// pattern-defeating quicksort, 2017..1.80) specialised to (key, id) records.
//
// Why the product needs this: the reference's kd-tree builder sorts a node's
// triangles by bbox-min on the split axis with sort_unstable_by
// (src/geometry/kdtree.rs:70).  Meshes exported from modelling tools share
// vertices, so equal keys are the norm, and where pdqsort happens to leave
// tied triangles decides which of them are duplicated into the left child
// (kdtree.rs:81-91) and in what order a leaf lists them -- which, through the
// traversal's first-hit-leaf rule (kdtree.rs:201-215), changes pixels.  With
// this order the 4K main.rs render is byte-identical to the reference's
// committed bruh.png; with a plain stable order 0.28 % of its pixels differ.
//
// The algorithm's published structure is followed stage by stage (pivot
// choice, already-sorted probe, equal-run partition, block partition,
// pattern breaking, heapsort fallback, insertion sort for <= 20); the code is
// written for this one record type, not transliterated.
#pragma once
#include <cstddef>
#include <cstdint>

namespace rrt {

struct SortRec {
    double key;
    int32_t id;
};

class UnstableSort {
  public:
    static void run(SortRec* v, size_t n) {
        unsigned budget = 0;
        for (size_t m = n; m != 0; m >>= 1) ++budget;  // floor(log2 n) + 1
        quick(v, n, nullptr, budget);
    }

  private:
    static bool lt(const SortRec& a, const SortRec& b) { return a.key < b.key; }
    static void swp(SortRec& a, SortRec& b) { SortRec t = a; a = b; b = t; }

    // move v[n-1] left to its place in the sorted prefix v[0..n-1)
    static void sink_last(SortRec* v, size_t n) {
        if (n < 2 || !lt(v[n - 1], v[n - 2])) return;
        const SortRec x = v[n - 1];
        size_t hole = n - 1;
        do {
            v[hole] = v[hole - 1];
            --hole;
        } while (hole > 0 && lt(x, v[hole - 1]));
        v[hole] = x;
    }
    // move v[0] right to its place in the sorted suffix v[1..n)
    static void float_first(SortRec* v, size_t n) {
        if (n < 2 || !lt(v[1], v[0])) return;
        const SortRec x = v[0];
        size_t hole = 0;
        do {
            v[hole] = v[hole + 1];
            ++hole;
        } while (hole + 1 < n && lt(v[hole + 1], x));
        v[hole] = x;
    }
    static void insertion(SortRec* v, size_t n) {
        for (size_t k = 2; k <= n; ++k) sink_last(v, k);
    }
    static bool nearly_sorted_fixup(SortRec* v, size_t n) {
        size_t i = 1;
        for (int step = 0; step < 5; ++step) {
            while (i < n && !lt(v[i], v[i - 1])) ++i;
            if (i == n) return true;
            if (n < 50) return false;
            swp(v[i - 1], v[i]);
            sink_last(v, i);
            float_first(v + i, n - i);
        }
        return false;
    }
    static void heap_sift(SortRec* v, size_t n, size_t at) {
        for (;;) {
            size_t kid = 2 * at + 1;
            if (kid + 1 < n && lt(v[kid], v[kid + 1])) ++kid;
            if (kid >= n || !lt(v[at], v[kid])) return;
            swp(v[at], v[kid]);
            at = kid;
        }
    }
    static void heap(SortRec* v, size_t n) {
        for (size_t i = n / 2; i > 0; --i) heap_sift(v, n, i - 1);
        for (size_t end = n; end > 1; --end) {
            swp(v[0], v[end - 1]);
            heap_sift(v, end - 1, 0);
        }
    }

    // Block partition of v[0..n) around `piv`: returns #elements < piv.
    static size_t block_partition(SortRec* v, size_t n, const SortRec& piv) {
        enum { B = 128 };
        uint8_t offL[B], offR[B];
        size_t lo = 0, hi = n;
        size_t widthL = B, widthR = B;
        size_t headL = 0, tailL = 0, headR = 0, tailR = 0;
        bool last = false;
        while (!last) {
            last = (hi - lo) <= 2 * B;
            if (last) {
                size_t rest = hi - lo;
                const bool pendL = headL < tailL, pendR = headR < tailR;
                if (pendL || pendR) rest -= B;
                if (pendL) widthR = rest;
                else if (pendR) widthL = rest;
                else { widthL = rest / 2; widthR = rest - widthL; }
            }
            if (headL == tailL) {
                headL = tailL = 0;
                for (size_t i = 0; i < widthL; ++i) {
                    offL[tailL] = (uint8_t)i;
                    tailL += lt(v[lo + i], piv) ? 0u : 1u;
                }
            }
            if (headR == tailR) {
                headR = tailR = 0;
                for (size_t i = 0; i < widthR; ++i) {
                    offR[tailR] = (uint8_t)i;
                    tailR += lt(v[hi - 1 - i], piv) ? 1u : 0u;
                }
            }
            const size_t nL = tailL - headL, nR = tailR - headR;
            const size_t m = nL < nR ? nL : nR;
            if (m > 0) {
                // rotate the m misplaced pairs as one cycle
                size_t a = lo + offL[headL], b = hi - 1 - offR[headR];
                const SortRec carry = v[a];
                v[a] = v[b];
                for (size_t k = 1; k < m; ++k) {
                    a = lo + offL[++headL];
                    v[b] = v[a];
                    b = hi - 1 - offR[++headR];
                    v[a] = v[b];
                }
                v[b] = carry;
                ++headL;
                ++headR;
            }
            if (headL == tailL) lo += widthL;
            if (headR == tailR) hi -= widthR;
        }
        if (headL < tailL) {
            while (headL < tailL) {
                --tailL;
                --hi;
                swp(v[lo + offL[tailL]], v[hi]);
            }
            return hi;
        }
        if (headR < tailR) {
            while (headR < tailR) {
                --tailR;
                swp(v[lo], v[hi - 1 - offR[tailR]]);
                ++lo;
            }
            return lo;
        }
        return lo;
    }

    // returns pivot's final index; *untouched = slice was already partitioned
    static size_t split(SortRec* v, size_t n, size_t pick, bool* untouched) {
        swp(v[0], v[pick]);
        const SortRec piv = v[0];
        size_t l = 1, r = n;
        while (l < r && lt(v[l], piv)) ++l;
        while (l < r && !lt(v[r - 1], piv)) --r;
        *untouched = l >= r;
        const size_t mid = (l - 1) + block_partition(v + l, r - l, piv);
        swp(v[0], v[mid]);
        return mid;
    }
    // slice has nothing below v[pick]; returns count of elements equal to it
    static size_t split_equal(SortRec* v, size_t n, size_t pick) {
        swp(v[0], v[pick]);
        const SortRec piv = v[0];
        size_t l = 1, r = n;
        for (;;) {
            while (l < r && !lt(piv, v[l])) ++l;
            while (l < r && lt(piv, v[r - 1])) --r;
            if (l >= r) break;
            --r;
            swp(v[l], v[r]);
            ++l;
        }
        return l;
    }
    static void scramble(SortRec* v, size_t n) {
        if (n < 8) return;
        uint32_t x = (uint32_t)n;
        auto next32 = [&x]() {
            x ^= x << 13;
            x ^= x >> 17;
            x ^= x << 5;
            return x;
        };
        size_t mask = 1;
        while (mask < n) mask <<= 1;
        --mask;
        const size_t around = n / 4 * 2;
        for (size_t i = 0; i < 3; ++i) {
            const uint64_t hi = next32();
            const uint64_t lo = next32();
            size_t j = (size_t)(((hi << 32) | lo) & (uint64_t)mask);
            if (j >= n) j -= n;
            swp(v[around - 1 + i], v[j]);
        }
    }
    static size_t pick_pivot(SortRec* v, size_t n, bool* looks_sorted) {
        size_t a = n / 4, b = n / 4 * 2, c = n / 4 * 3;
        unsigned flips = 0;
        auto order2 = [&](size_t& p, size_t& q) {
            if (lt(v[q], v[p])) { size_t t = p; p = q; q = t; ++flips; }
        };
        auto order3 = [&](size_t& p, size_t& q, size_t& r) { order2(p, q); order2(q, r); order2(p, q); };
        if (n >= 8) {
            if (n >= 50) {
                size_t* mids[3] = {&a, &b, &c};
                for (size_t* m : mids) {
                    size_t below = *m - 1, above = *m + 1;
                    order3(below, *m, above);
                }
            }
            order3(a, b, c);
        }
        if (flips < 12) {
            *looks_sorted = flips == 0;
            return b;
        }
        for (size_t i = 0, j = n - 1; i < j; ++i, --j) swp(v[i], v[j]);
        *looks_sorted = true;
        return n - 1 - b;
    }

    static void quick(SortRec* v, size_t n, const SortRec* floor_rec, unsigned budget) {
        bool balanced = true, untouched = true;
        SortRec floor_copy;
        for (;;) {
            if (n <= 20) { insertion(v, n); return; }
            if (budget == 0) { heap(v, n); return; }
            if (!balanced) { scramble(v, n); --budget; }
            bool looks_sorted;
            const size_t pick = pick_pivot(v, n, &looks_sorted);
            if (balanced && untouched && looks_sorted && nearly_sorted_fixup(v, n)) return;
            if (floor_rec && !lt(*floor_rec, v[pick])) {
                const size_t eq = split_equal(v, n, pick);
                v += eq;
                n -= eq;
                continue;
            }
            const size_t mid = split(v, n, pick, &untouched);
            const size_t nl = mid, nr = n - mid - 1;
            balanced = (nl < n - mid ? nl : n - mid) >= n / 8;
            const SortRec pivot = v[mid];
            if (nl < nr) {
                quick(v, nl, floor_rec, budget);
                floor_copy = pivot;
                floor_rec = &floor_copy;
                v += mid + 1;
                n = nr;
            } else {
                quick(v + mid + 1, nr, &pivot, budget);
                n = nl;
            }
        }
    }
};

}  // namespace rrt
