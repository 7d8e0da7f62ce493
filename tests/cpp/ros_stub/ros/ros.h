// Stub of the part of roscpp the node shims use -- for the COMPILE-ONLY test of ros/*.cpp (tests/test_ros_shims.py).
// Same names and signatures as roscpp; nothing here talks to a ROS master.
#pragma once
#include <cstdint>
#include <iostream>
#include <string>

namespace ros {
struct Duration {
    double s = 0;
    double toSec() const { return s; }
};
struct Time {
    uint32_t sec = 0, nsec = 0;
    static Time now() { return Time(); }
    Duration operator-(const Time &o) const { return Duration{(double)sec - (double)o.sec + 1e-9 * ((double)nsec - (double)o.nsec)}; }
};
struct Publisher {
    template <class M>
    void publish(const M &) const {}
};
struct Subscriber {
    ~Subscriber() {} // (the real one unsubscribes here)
};
struct Rate {
    explicit Rate(double) {}
    bool sleep() { return true; }
};
struct NodeHandle {
    template <class M>
    Subscriber subscribe(const std::string &, uint32_t, void (*)(const M &)) { return Subscriber(); }
    template <class M>
    Publisher advertise(const std::string &, uint32_t, bool = false) { return Publisher(); }
};
inline void init(int &, char **, const std::string &) {}
inline void spin() {}
inline void spinOnce() {}
inline bool ok() { return false; }
} // namespace ros
#define ROS_WARN_STREAM(x) (std::cerr << x << std::endl)
#define ROS_ERROR_STREAM(x) (std::cerr << x << std::endl)
