// DefaultInit.h — an allocator whose resize() leaves new elements uninitialised: for byte buffers that are written in
// full right after they are sized (zero-filling 300 MB first costs 50 ms on one thread and places every page on its node).
#ifndef HC_DEFAULT_INIT_H_
#define HC_DEFAULT_INIT_H_
#include <memory>
#include <new>
#include <type_traits>
#include <utility>

namespace hc {

template <class T>
struct DefaultInitAllocator : std::allocator<T> {
    template <class U>
    struct rebind {
        using other = DefaultInitAllocator<U>;
    };
    using std::allocator<T>::allocator;
    template <class U>
    void construct(U* p) noexcept(std::is_nothrow_default_constructible<U>::value) {
        ::new ((void*)p) U;
    }
    template <class U, class... A>
    void construct(U* p, A&&... a) {
        ::new ((void*)p) U(std::forward<A>(a)...);
    }
};

}  // namespace hc
#endif
