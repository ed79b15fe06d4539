// Minimal stand-in for <boost/multi_array.hpp> (oracle build only): the one
// 2-D array + boost::extents[a][b] form used by the reference's FDR compiler.
#pragma once
#include <cstddef>
#include <vector>
namespace boost {
namespace ma_detail {
struct extent2 { std::size_t a, b; };
struct extent1 {
    std::size_t a;
    extent2 operator[](std::size_t b) const { return extent2{a, b}; }
};
struct extent0 {
    extent1 operator[](std::size_t a) const { return extent1{a}; }
};
} // namespace ma_detail
static const ma_detail::extent0 extents = ma_detail::extent0();

template <class T, std::size_t N> class multi_array;
template <class T> class multi_array<T, 2> {
public:
    explicit multi_array(const ma_detail::extent2 &e) : cols(e.b), data(e.a * e.b) {}
    T *operator[](std::size_t i) { return data.data() + i * cols; }
    const T *operator[](std::size_t i) const { return data.data() + i * cols; }
private:
    std::size_t cols;
    std::vector<T> data;
};
} // namespace boost
