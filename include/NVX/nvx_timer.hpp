// NVX/nvx_timer.hpp -- nvx::Timer as used by demos/feature_tracker/main_feature_tracker.cpp:312-341
// (tic(); toc() returns milliseconds).  Header-only, host clock.
#ifndef NVX_TIMER_HPP
#define NVX_TIMER_HPP

#include <chrono>

namespace nvx {

class Timer {
public:
    Timer() { tic(); }
    void tic() { start_ = std::chrono::steady_clock::now(); }
    double toc() const
    {
        return std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - start_).count();
    }

private:
    std::chrono::steady_clock::time_point start_;
};

}  // namespace nvx

#endif
