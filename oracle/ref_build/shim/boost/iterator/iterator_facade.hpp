// Minimal stand-in for <boost/iterator/iterator_facade.hpp> (oracle build only):
// a CRTP random-access facade with by-value Reference support.
#pragma once
#include <cstddef>
#include <iterator>
#include <memory>
#include <type_traits>
namespace boost {
struct random_access_traversal_tag {};
struct bidirectional_traversal_tag {};
struct forward_traversal_tag {};

class iterator_core_access {
public:
    template <class F> static typename F::reference dereference(const F &f) { return f.dereference(); }
    template <class F> static void increment(F &f) { f.increment(); }
    template <class F> static void decrement(F &f) { f.decrement(); }
    template <class F, class D> static void advance(F &f, D n) { f.advance(n); }
    template <class F1, class F2> static bool equal(const F1 &a, const F2 &b) { return a.equal(b); }
    template <class F1, class F2>
    static auto distance_from(const F1 &a, const F2 &b) -> decltype(b.distance_to(a)) { return b.distance_to(a); }
};

namespace facade_detail {
template <class Ref> struct arrow {
    // Reference is a real reference: operator-> yields a pointer.
    typedef typename std::remove_reference<Ref>::type *type;
    static type make(Ref r) { return std::addressof(r); }
};
template <class V> struct arrow_proxy {
    V v;
    V *operator->() { return &v; }
};
template <class Ref, bool IsRef = std::is_reference<Ref>::value> struct arrow_sel;
template <class Ref> struct arrow_sel<Ref, true> {
    typedef typename std::remove_reference<Ref>::type *type;
    static type make(Ref r) { return std::addressof(r); }
};
template <class Ref> struct arrow_sel<Ref, false> {
    typedef arrow_proxy<typename std::remove_const<Ref>::type> type;
    static type make(const Ref &r) { return type{r}; }
};
} // namespace facade_detail

template <class Derived, class Value, class Traversal, class Reference = Value &,
          class Difference = std::ptrdiff_t>
class iterator_facade {
    Derived &derived() { return *static_cast<Derived *>(this); }
    const Derived &derived() const { return *static_cast<const Derived *>(this); }
public:
    typedef typename std::remove_const<Value>::type value_type;
    typedef Reference reference;
    typedef Difference difference_type;
    typedef typename facade_detail::arrow_sel<Reference>::type pointer;
    typedef std::random_access_iterator_tag iterator_category;

    reference operator*() const { return iterator_core_access::dereference(derived()); }
    pointer operator->() const { return facade_detail::arrow_sel<Reference>::make(*derived()); }
    reference operator[](difference_type n) const { Derived t(derived()); t += n; return *t; }
    Derived &operator++() { iterator_core_access::increment(derived()); return derived(); }
    Derived operator++(int) { Derived t(derived()); ++*this; return t; }
    Derived &operator--() { iterator_core_access::decrement(derived()); return derived(); }
    Derived operator--(int) { Derived t(derived()); --*this; return t; }
    Derived &operator+=(difference_type n) { iterator_core_access::advance(derived(), n); return derived(); }
    Derived &operator-=(difference_type n) { iterator_core_access::advance(derived(), -n); return derived(); }
    Derived operator+(difference_type n) const { Derived t(derived()); t += n; return t; }
    Derived operator-(difference_type n) const { Derived t(derived()); t -= n; return t; }
    friend Derived operator+(difference_type n, const Derived &d) { return d + n; }
};

#define HSREF_FACADE_TPL(n) class D##n, class V##n, class T##n, class R##n, class F##n
#define HSREF_FACADE(n) iterator_facade<D##n, V##n, T##n, R##n, F##n>
template <HSREF_FACADE_TPL(1), HSREF_FACADE_TPL(2)>
bool operator==(const HSREF_FACADE(1) &a, const HSREF_FACADE(2) &b) {
    return iterator_core_access::equal(static_cast<const D1 &>(a), static_cast<const D2 &>(b));
}
template <HSREF_FACADE_TPL(1), HSREF_FACADE_TPL(2)>
bool operator!=(const HSREF_FACADE(1) &a, const HSREF_FACADE(2) &b) { return !(a == b); }
template <HSREF_FACADE_TPL(1), HSREF_FACADE_TPL(2)>
auto operator-(const HSREF_FACADE(1) &a, const HSREF_FACADE(2) &b)
    -> decltype(iterator_core_access::distance_from(static_cast<const D1 &>(a), static_cast<const D2 &>(b))) {
    return iterator_core_access::distance_from(static_cast<const D1 &>(a), static_cast<const D2 &>(b));
}
template <HSREF_FACADE_TPL(1), HSREF_FACADE_TPL(2)>
bool operator<(const HSREF_FACADE(1) &a, const HSREF_FACADE(2) &b) { return (a - b) < 0; }
template <HSREF_FACADE_TPL(1), HSREF_FACADE_TPL(2)>
bool operator>(const HSREF_FACADE(1) &a, const HSREF_FACADE(2) &b) { return (a - b) > 0; }
template <HSREF_FACADE_TPL(1), HSREF_FACADE_TPL(2)>
bool operator<=(const HSREF_FACADE(1) &a, const HSREF_FACADE(2) &b) { return (a - b) <= 0; }
template <HSREF_FACADE_TPL(1), HSREF_FACADE_TPL(2)>
bool operator>=(const HSREF_FACADE(1) &a, const HSREF_FACADE(2) &b) { return (a - b) >= 0; }
#undef HSREF_FACADE_TPL
#undef HSREF_FACADE
} // namespace boost
