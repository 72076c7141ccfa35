// =============================================================================
// oracle/rust_sort.hpp -- TEST INFRASTRUCTURE (part of the parity oracle).
//
// Restatement of Rust's `slice::sort_unstable_by` as shipped in libcore from
// 2017 until 1.80 (pattern-defeating quicksort, library/core/src/slice/sort.rs;
// the reference's Cargo.lock dates the build to ~rustc 1.50, spring 2021).
// The reference's kd-tree builder calls it at src/geometry/kdtree.rs:70 and the
// model has many tied keys, so the *tie order* of this exact algorithm decides
// leaf membership.  libcore is not vendored in /root/reference;
// this is a restatement of its published algorithm: choose_pivot (median of
// three / medians of adjacent triples, >=12 swaps => reverse), partial
// insertion sort, partition_equal on pivot==predecessor, BlockQuicksort
// partition with 128-element blocks and cyclic permutation, break_patterns
// with a xorshift32 seeded by len, heapsort fallback after
// floor(log2(len))+1 imbalanced partitions, insertion sort for len<=20.
// It is deterministic.  Pinned only empirically: with it the oracle reproduces
// bruh.png's UFO pixels (tests/test_oracle_golden.py).
// =============================================================================
#pragma once
#include <algorithm>
#include <cstddef>
#include <cstdint>
#include <utility>

namespace rust_sort {

template <class T, class Less>
void shift_tail(T* v, size_t len, Less& is_less) {
    if (len >= 2 && is_less(v[len - 1], v[len - 2])) {
        T tmp = v[len - 1];
        v[len - 1] = v[len - 2];
        size_t hole = len - 2;
        for (size_t k = len - 2; k-- > 0;) {
            if (!is_less(tmp, v[k])) break;
            v[k + 1] = v[k];
            hole = k;
        }
        v[hole] = tmp;
    }
}

template <class T, class Less>
void shift_head(T* v, size_t len, Less& is_less) {
    if (len >= 2 && is_less(v[1], v[0])) {
        T tmp = v[0];
        v[0] = v[1];
        size_t hole = 1;
        for (size_t i = 2; i < len; ++i) {
            if (!is_less(v[i], tmp)) break;
            v[i - 1] = v[i];
            hole = i;
        }
        v[hole] = tmp;
    }
}

template <class T, class Less>
bool partial_insertion_sort(T* v, size_t len, Less& is_less) {
    const size_t MAX_STEPS = 5, SHORTEST_SHIFTING = 50;
    size_t i = 1;
    for (size_t step = 0; step < MAX_STEPS; ++step) {
        while (i < len && !is_less(v[i], v[i - 1])) ++i;
        if (i == len) return true;
        if (len < SHORTEST_SHIFTING) return false;
        std::swap(v[i - 1], v[i]);
        shift_tail(v, i, is_less);
        shift_head(v + i, len - i, is_less);
    }
    return false;
}

template <class T, class Less>
void insertion_sort(T* v, size_t len, Less& is_less) {
    for (size_t i = 1; i < len; ++i) shift_tail(v, i + 1, is_less);
}

template <class T, class Less>
void heapsort(T* v, size_t len, Less& is_less) {
    auto sift_down = [&](T* a, size_t n, size_t node) {
        for (;;) {
            const size_t left = 2 * node + 1, right = 2 * node + 2;
            const size_t greater = (right < n && is_less(a[left], a[right])) ? right : left;
            if (greater >= n || !is_less(a[node], a[greater])) break;
            std::swap(a[node], a[greater]);
            node = greater;
        }
    };
    for (size_t i = len / 2; i-- > 0;) sift_down(v, len, i);
    for (size_t i = len; i-- > 1;) {
        std::swap(v[0], v[i]);
        sift_down(v, i, 0);
    }
}

template <class T, class Less>
size_t partition_in_blocks(T* v, size_t n, const T& pivot, Less& is_less) {
    const size_t BLOCK = 128;
    size_t l = 0, block_l = BLOCK, start_l = 0, end_l = 0;
    size_t r = n, block_r = BLOCK, start_r = 0, end_r = 0;
    uint8_t off_l[BLOCK], off_r[BLOCK];
    for (;;) {
        const bool is_done = (r - l) <= 2 * BLOCK;
        if (is_done) {
            size_t rem = r - l;
            if (start_l < end_l || start_r < end_r) rem -= BLOCK;
            if (start_l < end_l) block_r = rem;
            else if (start_r < end_r) block_l = rem;
            else { block_l = rem / 2; block_r = rem - block_l; }
        }
        if (start_l == end_l) {
            start_l = end_l = 0;
            for (size_t i = 0; i < block_l; ++i) {
                off_l[end_l] = (uint8_t)i;
                end_l += is_less(v[l + i], pivot) ? 0 : 1;
            }
        }
        if (start_r == end_r) {
            start_r = end_r = 0;
            for (size_t i = 0; i < block_r; ++i) {
                off_r[end_r] = (uint8_t)i;
                end_r += is_less(v[r - 1 - i], pivot) ? 1 : 0;
            }
        }
        const size_t count = std::min(end_l - start_l, end_r - start_r);
        if (count > 0) {
            auto L = [&]() -> T& { return v[l + off_l[start_l]]; };
            auto R = [&]() -> T& { return v[r - off_r[start_r] - 1]; };
            T tmp = L();
            L() = R();
            for (size_t k = 1; k < count; ++k) {
                ++start_l;
                R() = L();
                ++start_r;
                L() = R();
            }
            R() = tmp;
            ++start_l;
            ++start_r;
        }
        if (start_l == end_l) l += block_l;
        if (start_r == end_r) r -= block_r;
        if (is_done) break;
    }
    if (start_l < end_l) {
        while (start_l < end_l) {
            --end_l;
            std::swap(v[l + off_l[end_l]], v[r - 1]);
            --r;
        }
        return r;
    } else if (start_r < end_r) {
        while (start_r < end_r) {
            --end_r;
            std::swap(v[l], v[r - off_r[end_r] - 1]);
            ++l;
        }
        return l;
    }
    return l;
}

template <class T, class Less>
std::pair<size_t, bool> partition(T* v, size_t len, size_t pivot_idx, Less& is_less) {
    std::swap(v[0], v[pivot_idx]);
    const T pivot = v[0];
    T* w = v + 1;
    size_t l = 0, r = len - 1;
    while (l < r && is_less(w[l], pivot)) ++l;
    while (l < r && !is_less(w[r - 1], pivot)) --r;
    const size_t mid = l + partition_in_blocks(w + l, r - l, pivot, is_less);
    const bool was_partitioned = l >= r;
    std::swap(v[0], v[mid]);
    return {mid, was_partitioned};
}

template <class T, class Less>
size_t partition_equal(T* v, size_t len, size_t pivot_idx, Less& is_less) {
    std::swap(v[0], v[pivot_idx]);
    const T pivot = v[0];
    T* w = v + 1;
    size_t l = 0, r = len - 1;
    for (;;) {
        while (l < r && !is_less(pivot, w[l])) ++l;
        while (l < r && is_less(pivot, w[r - 1])) --r;
        if (l >= r) break;
        --r;
        std::swap(w[l], w[r]);
        ++l;
    }
    return l + 1;
}

template <class T>
void break_patterns(T* v, size_t len) {
    if (len >= 8) {
        uint32_t random = (uint32_t)len;
        auto gen_u32 = [&]() {
            random ^= random << 13;
            random ^= random >> 17;
            random ^= random << 5;
            return random;
        };
        auto gen_usize = [&]() -> uint64_t {   // 64-bit usize
            const uint64_t hi = gen_u32();
            const uint64_t lo = gen_u32();
            return (hi << 32) | lo;
        };
        size_t modulus = 1;
        while (modulus < len) modulus <<= 1;  // next_power_of_two
        const size_t pos = len / 4 * 2;
        for (size_t i = 0; i < 3; ++i) {
            size_t other = (size_t)(gen_usize() & (uint64_t)(modulus - 1));
            if (other >= len) other -= len;
            std::swap(v[pos - 1 + i], v[other]);
        }
    }
}

template <class T, class Less>
std::pair<size_t, bool> choose_pivot(T* v, size_t len, Less& is_less) {
    const size_t SHORTEST_MEDIAN_OF_MEDIANS = 50, MAX_SWAPS = 4 * 3;
    size_t a = len / 4 * 1, b = len / 4 * 2, c = len / 4 * 3;
    size_t swaps = 0;
    if (len >= 8) {
        auto sort2 = [&](size_t& x, size_t& y) {
            if (is_less(v[y], v[x])) { std::swap(x, y); ++swaps; }
        };
        auto sort3 = [&](size_t& x, size_t& y, size_t& z) { sort2(x, y); sort2(y, z); sort2(x, y); };
        if (len >= SHORTEST_MEDIAN_OF_MEDIANS) {
            auto sort_adjacent = [&](size_t& x) {
                size_t lo = x - 1, hi = x + 1;
                sort3(lo, x, hi);
            };
            sort_adjacent(a);
            sort_adjacent(b);
            sort_adjacent(c);
        }
        sort3(a, b, c);
    }
    if (swaps < MAX_SWAPS) return {b, swaps == 0};
    std::reverse(v, v + len);
    return {len - 1 - b, true};
}

template <class T, class Less>
void recurse(T* v, size_t len, Less& is_less, const T* pred, uint32_t limit) {
    const size_t MAX_INSERTION = 20;
    bool was_balanced = true, was_partitioned = true;
    T pred_store{};
    for (;;) {
        if (len <= MAX_INSERTION) { insertion_sort(v, len, is_less); return; }
        if (limit == 0) { heapsort(v, len, is_less); return; }
        if (!was_balanced) { break_patterns(v, len); --limit; }
        auto [pivot, likely_sorted] = choose_pivot(v, len, is_less);
        if (was_balanced && was_partitioned && likely_sorted) {
            if (partial_insertion_sort(v, len, is_less)) return;
        }
        if (pred) {
            if (!is_less(*pred, v[pivot])) {
                const size_t mid = partition_equal(v, len, pivot, is_less);
                v += mid; len -= mid;
                continue;
            }
        }
        auto [mid, was_p] = partition(v, len, pivot, is_less);
        was_balanced = std::min(mid, len - mid) >= len / 8;
        was_partitioned = was_p;
        T* left = v; const size_t left_len = mid;
        T* right = v + mid + 1; const size_t right_len = len - mid - 1;
        const T pivot_val = v[mid];
        if (left_len < right_len) {
            recurse(left, left_len, is_less, pred, limit);
            v = right; len = right_len;
            pred_store = pivot_val; pred = &pred_store;
        } else {
            recurse(right, right_len, is_less, &pivot_val, limit);
            v = left; len = left_len;
        }
    }
}

// slice::sort_unstable_by(|a,b| ...) with is_less(a,b) == (compare(a,b) == Less)
template <class T, class Less>
void sort_unstable_by(T* v, size_t len, Less is_less) {
    uint32_t limit = 0;   // usize::BITS - len.leading_zeros()
    for (size_t n = len; n; n >>= 1) ++limit;
    recurse(v, len, is_less, (const T*)nullptr, limit);
}

}  // namespace rust_sort
