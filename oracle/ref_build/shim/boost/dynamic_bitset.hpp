// Minimal stand-in for <boost/dynamic_bitset.hpp> (oracle build only):
// just the members the reference's util/ headers touch.
#pragma once
#include <algorithm>
#include <cstddef>
#include <vector>
namespace boost {
template <class Block = unsigned long, class Alloc = void>
class dynamic_bitset {
public:
    typedef std::size_t size_type;
    typedef Block block_type;
    static const size_type npos = static_cast<size_type>(-1);
    static const size_type bits_per_block = sizeof(Block) * 8;

    class reference {
    public:
        reference(dynamic_bitset &b, size_type i) : bs(b), idx(i) {}
        operator bool() const { return bs.test(idx); }
        reference &operator=(bool v) { bs.set(idx, v); return *this; }
        reference &operator=(const reference &o) { bs.set(idx, bool(o)); return *this; }
    private:
        dynamic_bitset &bs;
        size_type idx;
    };

    dynamic_bitset() : nbits(0) {}
    explicit dynamic_bitset(size_type n, unsigned long value = 0)
        : blocks((n + bits_per_block - 1) / bits_per_block, 0), nbits(n) {
        if (!blocks.empty()) { blocks[0] = static_cast<Block>(value); trim(); }
    }
    size_type size() const { return nbits; }
    bool empty() const { return nbits == 0; }
    void clear() { blocks.clear(); nbits = 0; }
    void resize(size_type n, bool value = false) {
        size_type old = nbits;
        blocks.resize((n + bits_per_block - 1) / bits_per_block, 0);
        nbits = n;
        if (value) { for (size_type i = old; i < n; i++) set(i); }
        trim();
    }
    void push_back(bool b) { resize(nbits + 1); set(nbits - 1, b); }
    void swap(dynamic_bitset &o) { blocks.swap(o.blocks); std::swap(nbits, o.nbits); }
    dynamic_bitset &set(size_type i, bool v = true) {
        Block m = Block(1) << (i % bits_per_block);
        if (v) blocks[i / bits_per_block] |= m; else blocks[i / bits_per_block] &= ~m;
        return *this;
    }
    dynamic_bitset &set() { for (auto &b : blocks) b = ~Block(0); trim(); return *this; }
    dynamic_bitset &reset(size_type i) { return set(i, false); }
    dynamic_bitset &reset() { for (auto &b : blocks) b = 0; return *this; }
    dynamic_bitset &flip(size_type i) { blocks[i / bits_per_block] ^= Block(1) << (i % bits_per_block); return *this; }
    bool test(size_type i) const { return (blocks[i / bits_per_block] >> (i % bits_per_block)) & 1; }
    bool operator[](size_type i) const { return test(i); }
    reference operator[](size_type i) { return reference(*this, i); }
    bool any() const { for (auto b : blocks) if (b) return true; return false; }
    bool none() const { return !any(); }
    size_type count() const { size_type c = 0; for (auto b : blocks) c += __builtin_popcountl(b); return c; }
    size_type find_first() const { return scan_from(0); }
    size_type find_next(size_type i) const { return (i + 1 >= nbits) ? npos : scan_from(i + 1); }
    size_type num_blocks() const { return blocks.size(); }
    bool operator==(const dynamic_bitset &o) const { return nbits == o.nbits && blocks == o.blocks; }
    bool operator!=(const dynamic_bitset &o) const { return !(*this == o); }
    bool operator<(const dynamic_bitset &o) const {
        // numeric compare, most significant block first (sizes assumed equal as in boost)
        for (size_type i = blocks.size(); i-- > 0;) {
            Block a = blocks[i], b = i < o.blocks.size() ? o.blocks[i] : 0;
            if (a != b) return a < b;
        }
        return false;
    }
    template <class B, class A, class OutIt>
    friend void to_block_range(const dynamic_bitset<B, A> &bs, OutIt out);
private:
    size_type scan_from(size_type i) const {
        for (; i < nbits; i++) if (test(i)) return i;
        return npos;
    }
    void trim() {
        size_type extra = nbits % bits_per_block;
        if (extra && !blocks.empty()) blocks.back() &= (Block(1) << extra) - 1;
    }
    std::vector<Block> blocks;
    size_type nbits;
};
template <class B, class A, class OutIt>
void to_block_range(const dynamic_bitset<B, A> &bs, OutIt out) {
    for (auto b : bs.blocks) { *out = b; ++out; }
}
} // namespace boost
