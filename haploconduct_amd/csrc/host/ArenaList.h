// ArenaList.h — the adjacency "list" of the overlap graph: an order-preserving array with the members the path
// uses of the reference's std::list (push_back, erase, iteration, size), whose storage is either its own heap block
// or a slice of an arena shared by all lists of the graph.  The graph arrives from the device as CSR (one array of
// edges, one of offsets): every list then simply points at its stretch of that array — no allocation per vertex
// (one million small allocations cost more than scoring 10^8 candidates) — and turns into an owning list by itself
// the first time it has to grow.
#pragma once
#include <cstdlib>
#include <cstring>
#include <new>
#include <type_traits>

namespace hc {

template <typename T>
class ArenaList {
    static_assert(std::is_trivially_copyable<T>::value, "elements are moved with memcpy / memmove");

public:
    typedef T* iterator;
    typedef const T* const_iterator;
    typedef T value_type;

    ArenaList() = default;
    ArenaList(const ArenaList& o) { assign(o.m_data, o.m_size); }
    ArenaList(ArenaList&& o) noexcept : m_data(o.m_data), m_size(o.m_size), m_cap(o.m_cap), m_owned(o.m_owned) { o.forget(); }
    ArenaList& operator=(const ArenaList& o) {
        if (this != &o) {
            m_size = 0;
            assign(o.m_data, o.m_size);
        }
        return *this;
    }
    ArenaList& operator=(ArenaList&& o) noexcept {
        if (this != &o) {
            release();
            m_data = o.m_data, m_size = o.m_size, m_cap = o.m_cap, m_owned = o.m_owned;
            o.forget();
        }
        return *this;
    }
    ~ArenaList() { release(); }

    // point at n elements (room for cap >= n) of storage someone else owns and keeps alive
    void borrow(T* p, size_t n, size_t cap) {
        release();
        m_data = p;
        m_size = n;
        m_cap = cap;
        m_owned = false;
    }
    bool owns_storage() const { return m_owned; }

    iterator begin() { return m_data; }
    iterator end() { return m_data + m_size; }
    const_iterator begin() const { return m_data; }
    const_iterator end() const { return m_data + m_size; }
    size_t size() const { return m_size; }
    size_t capacity() const { return m_cap; }
    bool empty() const { return m_size == 0; }
    T* data() { return m_data; }
    const T* data() const { return m_data; }
    T& operator[](size_t i) { return m_data[i]; }
    const T& operator[](size_t i) const { return m_data[i]; }
    T& front() { return m_data[0]; }
    T& back() { return m_data[m_size - 1]; }
    const T& front() const { return m_data[0]; }
    const T& back() const { return m_data[m_size - 1]; }
    void clear() { m_size = 0; }  // keeps the storage, like std::vector
    void reserve(size_t n) {
        if (n > m_cap) grow(n);
    }
    void push_back(const T& v) {
        if (m_size == m_cap) {
            const T tmp = v;  // v may live in this list
            grow(m_cap < 4 ? 4 : 2 * m_cap);
            m_data[m_size++] = tmp;
            return;
        }
        m_data[m_size++] = v;
    }
    iterator erase(iterator it) {  // order-preserving, like std::list::erase
        memmove((void*)it, (const void*)(it + 1), (size_t)(end() - (it + 1)) * sizeof(T));
        m_size--;
        return it;
    }

private:
    void forget() {
        m_data = nullptr;
        m_size = m_cap = 0;
        m_owned = false;
    }
    void release() {
        if (m_owned) free((void*)m_data);
        forget();
    }
    void grow(size_t cap) {
        T* p = (T*)malloc(cap * sizeof(T));
        if (!p) throw std::bad_alloc();
        if (m_size) memcpy((void*)p, (const void*)m_data, m_size * sizeof(T));
        if (m_owned) free((void*)m_data);
        m_data = p;
        m_cap = cap;
        m_owned = true;
    }
    void assign(const T* p, size_t n) {
        if (n > m_cap) grow(n);
        if (n) memcpy((void*)m_data, (const void*)p, n * sizeof(T));
        m_size = n;
    }
    T* m_data = nullptr;
    size_t m_size = 0, m_cap = 0;
    bool m_owned = false;
};

}  // namespace hc
