// Minimal persistent worker pool for the host-side per-frame work of a batch (frames are independent).
#pragma once
#include <atomic>
#include <condition_variable>
#include <functional>
#include <mutex>
#include <thread>
#include <vector>

namespace vt
{

class Pool
{
public:
  explicit Pool(unsigned n_threads)
  {
    for (unsigned i = 0; i + 1 < n_threads; i++)  // the caller is the n-th worker
      workers_.emplace_back([this] { loop(); });
  }
  ~Pool()
  {
    {
      std::lock_guard<std::mutex> l(m_);
      stop_ = true;
    }
    cv_.notify_all();
    for (auto& t : workers_)
      t.join();
  }
  // runs fn(i) for i in [0, n); returns when all are done
  void parallel_for(uint32_t n, const std::function<void(uint32_t)>& fn)
  {
    if (workers_.empty() || n <= 1)
    {
      for (uint32_t i = 0; i < n; i++)
        fn(i);
      return;
    }
    {
      std::lock_guard<std::mutex> l(m_);
      fn_ = &fn;
      n_ = n;
      next_.store(0);
      pending_ = static_cast<int>(workers_.size());
      gen_++;
    }
    cv_.notify_all();
    work();
    std::unique_lock<std::mutex> l(m_);
    done_.wait(l, [this] { return pending_ == 0; });
    fn_ = nullptr;
  }

private:
  void work()
  {
    for (;;)
    {
      const uint32_t i = next_.fetch_add(1);
      if (i >= n_)
        break;
      (*fn_)(i);
    }
  }
  void loop()
  {
    uint64_t seen = 0;
    for (;;)
    {
      {
        std::unique_lock<std::mutex> l(m_);
        cv_.wait(l, [&] { return stop_ || gen_ != seen; });
        if (stop_)
          return;
        seen = gen_;
      }
      work();
      {
        std::lock_guard<std::mutex> l(m_);
        if (--pending_ == 0)
          done_.notify_one();
      }
    }
  }
  std::vector<std::thread> workers_;
  std::mutex m_;
  std::condition_variable cv_, done_;
  const std::function<void(uint32_t)>* fn_ = nullptr;
  uint32_t n_ = 0;
  std::atomic<uint32_t> next_{0};
  int pending_ = 0;
  uint64_t gen_ = 0;
  bool stop_ = false;
};

}  // namespace vt
