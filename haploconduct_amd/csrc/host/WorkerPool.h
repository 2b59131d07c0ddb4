// WorkerPool.h — a few worker threads that live as long as their owner.  The stage enters short parallel phases
// hundreds of times per file (two or three per block in the parser, one per block for the Edge build); starting
// threads for each of them costs more than the phases themselves.
#pragma once
#include <condition_variable>
#include <cstdint>
#include <functional>
#include <mutex>
#include <thread>
#include <vector>

namespace hc {

class WorkerPool {
public:
    WorkerPool(const WorkerPool&) = delete;
    WorkerPool& operator=(const WorkerPool&) = delete;
    // on_start: run by every worker before it waits for work (the stage binds its threads to the device's NUMA node)
    explicit WorkerPool(unsigned int workers, std::function<void()> on_start = nullptr) {
        for (unsigned int w = 0; w < workers; w++)
            m_threads.emplace_back([this, w, on_start] {
                if (on_start) on_start();
                loop(w + 1);
            });
    }
    ~WorkerPool() {
        {
            std::lock_guard<std::mutex> g(m_mu);
            m_stop = true;
            m_generation++;
        }
        m_cv.notify_all();
        for (auto& t : m_threads) t.join();
    }
    unsigned int workers() const { return (unsigned int)m_threads.size(); }
    // fn(t) for t in [0, n): t = 0 on the caller, the rest on the workers (n - 1 <= workers())
    void run(unsigned int n, const std::function<void(unsigned int)>& fn) {
        if (n <= 1) {
            fn(0);
            return;
        }
        {
            std::lock_guard<std::mutex> g(m_mu);
            m_fn = &fn;
            m_n = n;
            m_pending = n - 1;
            m_generation++;
        }
        m_cv.notify_all();
        fn(0);
        std::unique_lock<std::mutex> g(m_mu);
        m_done.wait(g, [this] { return m_pending == 0; });
        m_fn = nullptr;
    }

private:
    void loop(unsigned int id) {
        uint64_t seen = 0;
        for (;;) {
            const std::function<void(unsigned int)>* fn = nullptr;
            {
                std::unique_lock<std::mutex> g(m_mu);
                m_cv.wait(g, [&] { return m_generation != seen; });
                seen = m_generation;
                if (m_stop) return;
                if (id < m_n) fn = m_fn;
            }
            if (fn) {
                (*fn)(id);
                std::lock_guard<std::mutex> g(m_mu);
                if (--m_pending == 0) m_done.notify_one();
            }
        }
    }
    std::vector<std::thread> m_threads;
    std::mutex m_mu;
    std::condition_variable m_cv, m_done;
    const std::function<void(unsigned int)>* m_fn = nullptr;
    unsigned int m_n = 0, m_pending = 0;
    uint64_t m_generation = 0;
    bool m_stop = false;
};

}  // namespace hc
