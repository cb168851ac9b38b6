// tests/simt/include/rocprim/rocprim.hpp - TEST INFRASTRUCTURE (tests/simt/simt.h): the rocPRIM device-wide primitives the
// library's load path calls (hb_plan.hip, hb_ingest.hip), restated on host memory with the standard library - same argument
// lists, same two-call protocol (a null temporary-storage pointer asks for its size), same results: stable least-significant-
// digit semantics on the bit range for the radix sorts, first-of-run for unique, order-preserving select.
#pragma once
#include <hip/hip_runtime.h>

#include <algorithm>
#include <numeric>
#include <type_traits>
#include <vector>

namespace rocprim {
using uint128_t = unsigned __int128;

template <class T>
struct plus {
    T operator()(const T &a, const T &b) const { return a + b; }
};
template <class T>
struct maximum {
    T operator()(const T &a, const T &b) const { return a < b ? b : a; }
};
template <class T>
struct equal_to {
    bool operator()(const T &a, const T &b) const { return a == b; }
};

template <class T>
class double_buffer {
public:
    double_buffer(T *current, T *alternate) : buf_{current, alternate}, sel_(0) {}
    T *current() const { return buf_[sel_]; }
    T *alternate() const { return buf_[sel_ ^ 1]; }
    void swap() { sel_ ^= 1; }

private:
    T *buf_[2];
    int sel_;
};

template <class T>
class counting_iterator {
public:
    using value_type = T;
    explicit counting_iterator(T v) : v_(v) {}
    T operator*() const { return v_; }
    T operator[](size_t i) const { return (T)(v_ + (T)i); }
    counting_iterator operator+(size_t i) const { return counting_iterator((T)(v_ + (T)i)); }

private:
    T v_;
};
template <class T>
counting_iterator<T> make_counting_iterator(T v) { return counting_iterator<T>(v); }

template <class It, class F>
class transform_iterator {
public:
    using value_type = std::decay_t<decltype(std::declval<F>()(std::declval<It>()[0]))>;
    transform_iterator(It it, F f) : it_(it), f_(f) {}
    value_type operator*() const { return f_(it_[0]); }
    value_type operator[](size_t i) const { return f_(it_[i]); }
    transform_iterator operator+(size_t i) const { return transform_iterator(it_ + i, f_); }

private:
    It it_;
    F f_;
};
template <class It, class F>
transform_iterator<It, F> make_transform_iterator(It it, F f) { return transform_iterator<It, F>(it, f); }

namespace detail {
inline bool size_query(void *tmp, size_t &bytes)
{
    if (tmp) return false;
    bytes = 256;
    return true;
}
template <class K>
inline K digit(K k, unsigned begin_bit, unsigned end_bit)
{
    const unsigned total = sizeof(K) * 8;
    if (end_bit > total) end_bit = total;
    if (begin_bit >= end_bit) return (K)0;
    K v = k >> begin_bit;
    const unsigned width = end_bit - begin_bit;
    if (width < total) v &= (((K)1) << width) - (K)1;
    return v;
}
template <class K>
std::vector<size_t> sorted_order(const K *keys, size_t n, unsigned begin_bit, unsigned end_bit)
{
    std::vector<size_t> order(n);
    std::iota(order.begin(), order.end(), (size_t)0);
    std::stable_sort(order.begin(), order.end(), [&](size_t a, size_t b) { return digit(keys[a], begin_bit, end_bit) < digit(keys[b], begin_bit, end_bit); });
    return order;
}
} // namespace detail

template <class In, class Out, class Init, class Op>
hipError_t exclusive_scan(void *tmp, size_t &bytes, In in, Out out, Init init, size_t n, Op op, hipStream_t = nullptr, bool = false)
{
    if (detail::size_query(tmp, bytes)) return hipSuccess;
    Init acc = init;
    for (size_t i = 0; i < n; i++) { // (in and out may alias)
        const Init v = (Init)in[i];
        out[i] = acc;
        acc = op(acc, v);
    }
    return hipSuccess;
}
template <class In, class Out, class Op>
hipError_t inclusive_scan(void *tmp, size_t &bytes, In in, Out out, size_t n, Op op, hipStream_t = nullptr, bool = false)
{
    if (detail::size_query(tmp, bytes)) return hipSuccess;
    using T = std::decay_t<decltype(out[0])>;
    T acc{};
    for (size_t i = 0; i < n; i++) {
        acc = i ? (T)op(acc, (T)in[i]) : (T)in[i];
        out[i] = acc;
    }
    return hipSuccess;
}
template <class In, class Out, class Init, class Op>
hipError_t reduce(void *tmp, size_t &bytes, In in, Out out, Init init, size_t n, Op op, hipStream_t = nullptr, bool = false)
{
    if (detail::size_query(tmp, bytes)) return hipSuccess;
    Init acc = init;
    for (size_t i = 0; i < n; i++) acc = op(acc, (Init)in[i]);
    *out = acc;
    return hipSuccess;
}
template <class K>
hipError_t radix_sort_keys(void *tmp, size_t &bytes, const K *in, K *out, size_t n, unsigned begin_bit = 0, unsigned end_bit = 8 * sizeof(K), hipStream_t = nullptr,
                           bool = false)
{
    if (detail::size_query(tmp, bytes)) return hipSuccess;
    const std::vector<size_t> order = detail::sorted_order(in, n, begin_bit, end_bit);
    std::vector<K> r(n);
    for (size_t i = 0; i < n; i++) r[i] = in[order[i]];
    std::copy(r.begin(), r.end(), out);
    return hipSuccess;
}
template <class K>
hipError_t radix_sort_keys(void *tmp, size_t &bytes, double_buffer<K> &keys, size_t n, unsigned begin_bit = 0, unsigned end_bit = 8 * sizeof(K), hipStream_t = nullptr,
                           bool = false)
{
    if (detail::size_query(tmp, bytes)) return hipSuccess;
    hipError_t e = radix_sort_keys(tmp, bytes, (const K *)keys.current(), keys.alternate(), n, begin_bit, end_bit);
    keys.swap();
    return e;
}
template <class K, class VIn, class V>
hipError_t radix_sort_pairs(void *tmp, size_t &bytes, const K *kin, K *kout, VIn vin, V *vout, size_t n, unsigned begin_bit = 0, unsigned end_bit = 8 * sizeof(K),
                            hipStream_t = nullptr, bool = false)
{
    if (detail::size_query(tmp, bytes)) return hipSuccess;
    const std::vector<size_t> order = detail::sorted_order(kin, n, begin_bit, end_bit);
    std::vector<K> rk(n);
    std::vector<V> rv(n);
    for (size_t i = 0; i < n; i++) {
        rk[i] = kin[order[i]];
        rv[i] = (V)vin[order[i]];
    }
    std::copy(rk.begin(), rk.end(), kout);
    std::copy(rv.begin(), rv.end(), vout);
    return hipSuccess;
}
template <class K, class V>
hipError_t radix_sort_pairs(void *tmp, size_t &bytes, double_buffer<K> &keys, double_buffer<V> &vals, size_t n, unsigned begin_bit = 0, unsigned end_bit = 8 * sizeof(K),
                            hipStream_t = nullptr, bool = false)
{
    if (detail::size_query(tmp, bytes)) return hipSuccess;
    hipError_t e = radix_sort_pairs(tmp, bytes, (const K *)keys.current(), keys.alternate(), (const V *)vals.current(), vals.alternate(), n, begin_bit, end_bit);
    keys.swap();
    vals.swap();
    return e;
}
// select by flags: out = the in[i] whose flag is non-zero, in order; *count = how many
template <class In, class Flags, class Out, class Count>
hipError_t select(void *tmp, size_t &bytes, In in, Flags flags, Out out, Count count, size_t n, hipStream_t = nullptr, bool = false)
{
    if (detail::size_query(tmp, bytes)) return hipSuccess;
    size_t k = 0;
    for (size_t i = 0; i < n; i++)
        if (flags[i]) out[k++] = in[i];
    *count = (std::decay_t<decltype(*count)>)k;
    return hipSuccess;
}
template <class In, class Out, class Count, class Eq>
hipError_t unique(void *tmp, size_t &bytes, In in, Out out, Count count, size_t n, Eq eq, hipStream_t = nullptr, bool = false)
{
    if (detail::size_query(tmp, bytes)) return hipSuccess;
    size_t k = 0;
    for (size_t i = 0; i < n; i++)
        if (i == 0 || !eq(in[i - 1], in[i])) out[k++] = in[i];
    *count = (std::decay_t<decltype(*count)>)k;
    return hipSuccess;
}
} // namespace rocprim
