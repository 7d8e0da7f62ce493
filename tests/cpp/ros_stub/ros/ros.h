// A single-process stand-in for the part of roscpp the node shims use (tests/test_ros_shims.py): same names and
// signatures as roscpp -- NodeHandle::subscribe / advertise, Publisher::publish, spin / spinOnce / ok, Rate, Time --
// with an in-process "master" behind them, so that the UNMODIFIED node sources (ros/*.cpp, their main() included) can be
// compiled and RUN by a harness without a ROS installation:
//   * a message published on a topic is recorded (ros::testing::published<M>(topic)) and queued for every subscriber of
//     that topic, newest `queue_size` kept per subscriber, delivered in arrival order by spinOnce() -- the callback-queue
//     behaviour the nodes rely on (all their subscriptions have queue size 1);
//   * where the node would block -- spin() with nothing queued, Rate::sleep() -- the harness's idle hook runs
//     (ros::testing::master().idle): it injects the next inputs (ros::testing::inject) or returns false, which is
//     ros::shutdown().  With no hook installed a node's main() returns at once.
// Nothing here talks to a ROS master, and none of it is part of the product.
#pragma once
#include <cstdint>
#include <deque>
#include <functional>
#include <iostream>
#include <memory>
#include <sstream>
#include <string>
#include <typeinfo>
#include <vector>

namespace ros {
struct Duration {
    double s = 0;
    double toSec() const { return s; }
};

namespace testing {
struct Sub {
    std::string           topic;
    uint32_t              queue_size = 1;
    const std::type_info *type = nullptr;
    std::function<void(const std::shared_ptr<const void> &)> deliver;
    bool                  alive = true;
};
struct Record {
    std::string                 topic;
    const std::type_info       *type;
    std::shared_ptr<const void> msg;
};
struct Master {
    std::vector<std::shared_ptr<Sub>> subs;
    std::deque<std::pair<std::shared_ptr<Sub>, std::shared_ptr<const void>>> queue; // arrival order
    std::vector<Record>   published;
    std::function<bool()> idle;     // the harness: inject inputs, return true; false = shut down
    bool                  shutdown = false;
    uint64_t              now_ns = 1000000000ull;
    std::vector<std::string> warnings, errors; // ROS_WARN_STREAM / ROS_ERROR_STREAM texts
};
inline Master &master()
{
    static Master m;
    return m;
}
inline std::string resolve(const std::string &name) { return (!name.empty() && name[0] == '/') ? name : "/" + name; } // (root namespace)

template <class M>
inline void deliver_to_subscribers(const std::string &topic, const std::shared_ptr<const M> &msg)
{
    Master &m = master();
    for (auto &s : m.subs) {
        if (!s->alive || s->topic != topic || *s->type != typeid(M)) continue;
        uint32_t pending = 0;
        for (auto &q : m.queue) pending += q.first == s ? 1u : 0u;
        while (pending >= s->queue_size && pending > 0) { // the subscriber's queue is full: its oldest message goes
            for (auto it = m.queue.begin(); it != m.queue.end(); ++it)
                if (it->first == s) {
                    m.queue.erase(it);
                    break;
                }
            --pending;
        }
        m.queue.emplace_back(s, std::static_pointer_cast<const void>(msg));
    }
}
// a message arriving from outside the node (what another node's publisher would send)
template <class M>
inline void inject(const std::string &topic, const M &msg)
{
    deliver_to_subscribers<M>(resolve(topic), std::make_shared<const M>(msg));
}
// everything the node published on `topic`, in order
template <class M>
inline std::vector<std::shared_ptr<const M>> published(const std::string &topic)
{
    std::vector<std::shared_ptr<const M>> out;
    for (auto &r : master().published)
        if (r.topic == resolve(topic) && *r.type == typeid(M)) out.push_back(std::static_pointer_cast<const M>(r.msg));
    return out;
}
inline bool run_idle()
{
    Master &m = master();
    if (!m.idle || !m.idle()) m.shutdown = true;
    return !m.shutdown;
}
} // namespace testing

struct Time {
    uint32_t sec = 0, nsec = 0;
    static Time now()
    {
        testing::Master &m = testing::master();
        m.now_ns += 1000000ull; // a millisecond per look: time passes
        Time t;
        t.sec = (uint32_t)(m.now_ns / 1000000000ull);
        t.nsec = (uint32_t)(m.now_ns % 1000000000ull);
        return t;
    }
    Duration operator-(const Time &o) const { return Duration{(double)sec - (double)o.sec + 1e-9 * ((double)nsec - (double)o.nsec)}; }
};

struct Publisher {
    std::string topic;
    template <class M>
    void publish(const M &msg) const
    {
        auto copy = std::make_shared<const M>(msg);
        testing::master().published.push_back({topic, &typeid(M), std::static_pointer_cast<const void>(copy)});
        testing::deliver_to_subscribers<M>(topic, copy);
    }
};

struct Subscriber {
    std::shared_ptr<testing::Sub> sub;
    Subscriber() = default;
    explicit Subscriber(std::shared_ptr<testing::Sub> s) : sub(std::move(s)) {}
    Subscriber(Subscriber &&) = default;
    Subscriber &operator=(Subscriber &&) = default;
    Subscriber(const Subscriber &) = delete;
    Subscriber &operator=(const Subscriber &) = delete;
    ~Subscriber() // (the real one unsubscribes here)
    {
        if (sub) sub->alive = false;
    }
};

struct NodeHandle {
    // callbacks taking the message by const reference ...
    template <class M>
    Subscriber subscribe(const std::string &topic, uint32_t queue_size, void (*cb)(const M &))
    {
        auto s = std::make_shared<testing::Sub>();
        s->topic = testing::resolve(topic);
        s->queue_size = queue_size ? queue_size : 1000000u; // 0 = unbounded in roscpp
        s->type = &typeid(M);
        s->deliver = [cb](const std::shared_ptr<const void> &m) { cb(*std::static_pointer_cast<const M>(m)); };
        testing::master().subs.push_back(s);
        return Subscriber(s);
    }
    // ... and by shared pointer to const (sensor_msgs::PointCloud2ConstPtr): the more specialised overload
    template <class M>
    Subscriber subscribe(const std::string &topic, uint32_t queue_size, void (*cb)(const std::shared_ptr<M const> &))
    {
        auto s = std::make_shared<testing::Sub>();
        s->topic = testing::resolve(topic);
        s->queue_size = queue_size ? queue_size : 1000000u;
        s->type = &typeid(M);
        s->deliver = [cb](const std::shared_ptr<const void> &m) { cb(std::static_pointer_cast<const M>(m)); };
        testing::master().subs.push_back(s);
        return Subscriber(s);
    }
    template <class M>
    Publisher advertise(const std::string &topic, uint32_t /*queue_size*/, bool /*latch*/ = false)
    {
        return Publisher{testing::resolve(topic)};
    }
};

inline void init(int &, char **, const std::string &) {}
inline bool ok() { return !testing::master().shutdown; }
inline void shutdown() { testing::master().shutdown = true; }
inline void spinOnce()
{
    testing::Master &m = testing::master();
    size_t n = m.queue.size(); // what has arrived by now (callbacks may publish: that is for the next round)
    while (n-- && !m.queue.empty()) {
        auto item = m.queue.front();
        m.queue.pop_front();
        if (item.first->alive) item.first->deliver(item.second);
    }
}
inline void spin()
{
    while (ok()) {
        spinOnce();
        if (testing::master().queue.empty()) testing::run_idle();
    }
}
struct Rate {
    double hz;
    explicit Rate(double f) : hz(f) {}
    bool sleep()
    {
        testing::Master &m = testing::master();
        m.now_ns += (uint64_t)(1e9 / (hz > 0 ? hz : 1.0));
        if (m.queue.empty()) testing::run_idle();
        return true;
    }
};
} // namespace ros
#define ROS_WARN_STREAM(x)                                                    \
    do {                                                                      \
        std::ostringstream ros_stub_os__;                                     \
        ros_stub_os__ << x;                                                   \
        ros::testing::master().warnings.push_back(ros_stub_os__.str());       \
        std::cerr << "[WARN] " << ros_stub_os__.str() << std::endl;           \
    } while (0)
#define ROS_ERROR_STREAM(x)                                                   \
    do {                                                                      \
        std::ostringstream ros_stub_os__;                                     \
        ros_stub_os__ << x;                                                   \
        ros::testing::master().errors.push_back(ros_stub_os__.str());         \
        std::cerr << "[ERROR] " << ros_stub_os__.str() << std::endl;          \
    } while (0)
