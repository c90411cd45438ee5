#include "store.h"

#include <arpa/inet.h>
#include <errno.h>
#include <netdb.h>
#include <netinet/in.h>
#include <netinet/tcp.h>
#include <poll.h>
#include <sys/socket.h>
#include <unistd.h>

#include <atomic>
#include <chrono>
#include <condition_variable>
#include <cstring>
#include <map>
#include <mutex>
#include <thread>

namespace ddl {

namespace {
enum Cmd : uint8_t { SET = 1, GET = 2, ADD = 3, DEL = 4 };

int64_t now_ms() {
    return std::chrono::duration_cast<std::chrono::milliseconds>(
               std::chrono::steady_clock::now().time_since_epoch()).count();
}
void set_nodelay(int fd) {
    int one = 1;
    setsockopt(fd, IPPROTO_TCP, TCP_NODELAY, &one, sizeof(one));
}
}  // namespace

bool send_all(int fd, const void* buf, size_t n) {
    const char* p = static_cast<const char*>(buf);
    while (n > 0) {
        ssize_t k = ::send(fd, p, n, MSG_NOSIGNAL);
        if (k < 0) {
            if (errno == EINTR) continue;
            if (errno == EAGAIN || errno == EWOULDBLOCK) {
                pollfd pf{fd, POLLOUT, 0};
                ::poll(&pf, 1, 1000);
                continue;
            }
            return false;
        }
        p += k; n -= (size_t)k;
    }
    return true;
}

bool recv_all(int fd, void* buf, size_t n, int timeout_ms) {
    char* p = static_cast<char*>(buf);
    const int64_t deadline = timeout_ms < 0 ? -1 : now_ms() + timeout_ms;
    while (n > 0) {
        if (deadline >= 0) {
            int64_t left = deadline - now_ms();
            if (left <= 0) return false;
            pollfd pf{fd, POLLIN, 0};
            int r = ::poll(&pf, 1, (int)(left > 1000 ? 1000 : left));
            if (r < 0 && errno != EINTR) return false;
            if (r <= 0) continue;
        }
        ssize_t k = ::recv(fd, p, n, 0);
        if (k == 0) return false;
        if (k < 0) {
            if (errno == EINTR || errno == EAGAIN || errno == EWOULDBLOCK) continue;
            return false;
        }
        p += k; n -= (size_t)k;
    }
    return true;
}

int tcp_listen(int port, int* bound_port, bool loopback_only) {
    int fd = ::socket(AF_INET, SOCK_STREAM, 0);
    if (fd < 0) return -1;
    int one = 1;
    setsockopt(fd, SOL_SOCKET, SO_REUSEADDR, &one, sizeof(one));
    sockaddr_in addr{};
    addr.sin_family = AF_INET;
    addr.sin_addr.s_addr = htonl(loopback_only ? INADDR_LOOPBACK : INADDR_ANY);
    addr.sin_port = htons((uint16_t)port);
    if (::bind(fd, (sockaddr*)&addr, sizeof(addr)) != 0 || ::listen(fd, 256) != 0) {
        ::close(fd);
        return -1;
    }
    socklen_t len = sizeof(addr);
    ::getsockname(fd, (sockaddr*)&addr, &len);
    if (bound_port) *bound_port = ntohs(addr.sin_port);
    return fd;
}

int tcp_connect(const std::string& host, int port, int timeout_ms) {
    const int64_t deadline = now_ms() + timeout_ms;
    addrinfo hints{}, *res = nullptr;
    hints.ai_family = AF_INET;
    hints.ai_socktype = SOCK_STREAM;
    const std::string ports = std::to_string(port);
    do {
        if (::getaddrinfo(host.c_str(), ports.c_str(), &hints, &res) == 0 && res) {
            int fd = ::socket(res->ai_family, res->ai_socktype, res->ai_protocol);
            if (fd >= 0) {
                if (::connect(fd, res->ai_addr, res->ai_addrlen) == 0) {
                    ::freeaddrinfo(res);
                    set_nodelay(fd);
                    return fd;
                }
                ::close(fd);
            }
            ::freeaddrinfo(res);
            res = nullptr;
        }
        std::this_thread::sleep_for(std::chrono::milliseconds(50));
    } while (now_ms() < deadline);
    return -1;
}

int tcp_accept(int listen_fd, int timeout_ms) {
    pollfd pf{listen_fd, POLLIN, 0};
    int r = ::poll(&pf, 1, timeout_ms);
    if (r <= 0) return -1;
    int fd = ::accept(listen_fd, nullptr, nullptr);
    if (fd >= 0) set_nodelay(fd);
    return fd;
}

// ------------------------------------------------------------------------------------------
class StoreServer {
public:
    int listen_fd = -1;
    std::atomic<bool> stop{false};
    std::thread acceptor;
    std::vector<std::thread> workers;
    std::vector<int> fds;
    std::mutex mu;
    std::condition_variable cv;
    std::map<std::string, std::string> kv;

    void serve(int fd) {
        for (;;) {
            uint8_t cmd;
            uint32_t klen;
            if (!recv_all(fd, &cmd, 1) || !recv_all(fd, &klen, 4) || klen > (1u << 20)) break;
            std::string key(klen, '\0');
            if (klen && !recv_all(fd, &key[0], klen)) break;
            if (cmd == SET) {
                uint64_t vlen;
                if (!recv_all(fd, &vlen, 8) || vlen > (1ull << 30)) break;
                std::string val(vlen, '\0');
                if (vlen && !recv_all(fd, &val[0], vlen)) break;
                {
                    std::lock_guard<std::mutex> g(mu);
                    kv[key] = std::move(val);
                }
                cv.notify_all();
                uint8_t okb = 1;
                if (!send_all(fd, &okb, 1)) break;
            } else if (cmd == GET) {
                int32_t timeout_ms;
                if (!recv_all(fd, &timeout_ms, 4)) break;
                std::string val;
                uint8_t found = 0;
                {
                    std::unique_lock<std::mutex> lk(mu);
                    auto pred = [&] { return stop.load() || kv.count(key) > 0; };
                    if (timeout_ms < 0) cv.wait(lk, pred);
                    else cv.wait_for(lk, std::chrono::milliseconds(timeout_ms), pred);
                    auto it = kv.find(key);
                    if (it != kv.end()) { val = it->second; found = 1; }
                }
                uint64_t vlen = val.size();
                if (!send_all(fd, &found, 1) || !send_all(fd, &vlen, 8) ||
                    (vlen && !send_all(fd, val.data(), vlen))) break;
            } else if (cmd == ADD) {
                int64_t delta, result;
                if (!recv_all(fd, &delta, 8)) break;
                {
                    std::lock_guard<std::mutex> g(mu);
                    int64_t cur = 0;
                    auto it = kv.find(key);
                    if (it != kv.end() && it->second.size() == 8) std::memcpy(&cur, it->second.data(), 8);
                    result = cur + delta;
                    kv[key] = std::string(reinterpret_cast<const char*>(&result), 8);
                }
                cv.notify_all();
                if (!send_all(fd, &result, 8)) break;
            } else if (cmd == DEL) {
                {
                    std::lock_guard<std::mutex> g(mu);
                    kv.erase(key);
                }
                uint8_t okb = 1;
                if (!send_all(fd, &okb, 1)) break;
            } else {
                break;
            }
        }
        ::close(fd);
    }

    void accept_loop() {
        while (!stop.load()) {
            int fd = tcp_accept(listen_fd, 200);
            if (fd < 0) continue;
            std::lock_guard<std::mutex> g(mu);
            fds.push_back(fd);
            workers.emplace_back([this, fd] { serve(fd); });
        }
    }
};

StoreServer* store_server_start(int port, int* bound_port, bool loopback_only) {
    auto* s = new StoreServer();
    s->listen_fd = tcp_listen(port, bound_port, loopback_only);
    if (s->listen_fd < 0) {
        delete s;
        return nullptr;
    }
    s->acceptor = std::thread([s] { s->accept_loop(); });
    return s;
}

void store_server_stop(StoreServer* s) {
    if (!s) return;
    s->stop.store(true);
    s->cv.notify_all();
    if (s->acceptor.joinable()) s->acceptor.join();
    {
        std::lock_guard<std::mutex> g(s->mu);
        for (int fd : s->fds) ::shutdown(fd, SHUT_RDWR);
    }
    for (auto& t : s->workers)
        if (t.joinable()) t.join();
    ::close(s->listen_fd);
    delete s;
}

// ------------------------------------------------------------------------------------------
StoreClient::~StoreClient() { close_fd(); }

void StoreClient::close_fd() {
    if (fd_ >= 0) {
        ::close(fd_);
        fd_ = -1;
    }
}

bool StoreClient::connect_to(const std::string& host, int port, int timeout_ms) {
    fd_ = tcp_connect(host, port, timeout_ms);
    return fd_ >= 0;
}

static bool send_head(int fd, uint8_t cmd, const std::string& key) {
    uint32_t klen = (uint32_t)key.size();
    return send_all(fd, &cmd, 1) && send_all(fd, &klen, 4) && (klen == 0 || send_all(fd, key.data(), klen));
}

bool StoreClient::set(const std::string& key, const std::string& value) {
    uint64_t vlen = value.size();
    uint8_t okb = 0;
    return fd_ >= 0 && send_head(fd_, SET, key) && send_all(fd_, &vlen, 8) &&
           (vlen == 0 || send_all(fd_, value.data(), vlen)) && recv_all(fd_, &okb, 1) && okb == 1;
}

bool StoreClient::get(const std::string& key, std::string* value, int timeout_ms) {
    int32_t t = timeout_ms;
    uint8_t found = 0;
    uint64_t vlen = 0;
    if (fd_ < 0 || !send_head(fd_, GET, key) || !send_all(fd_, &t, 4)) return false;
    if (!recv_all(fd_, &found, 1, timeout_ms < 0 ? -1 : timeout_ms + 5000) || !recv_all(fd_, &vlen, 8)) return false;
    value->assign(vlen, '\0');
    if (vlen && !recv_all(fd_, &(*value)[0], vlen)) return false;
    return found == 1;
}

bool StoreClient::add(const std::string& key, int64_t delta, int64_t* result) {
    return fd_ >= 0 && send_head(fd_, ADD, key) && send_all(fd_, &delta, 8) && recv_all(fd_, result, 8);
}

bool StoreClient::del(const std::string& key) {
    uint8_t okb = 0;
    return fd_ >= 0 && send_head(fd_, DEL, key) && recv_all(fd_, &okb, 1) && okb == 1;
}

}  // namespace ddl
