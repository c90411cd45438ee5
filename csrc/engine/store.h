// TCP key-value rendezvous store: the control plane that replaces MPI in this stack
// (SURVEY §2.2 N4, §5.8 "Control plane").  Rank 0 hosts the server; every rank is a client.
#pragma once
#include <cstdint>
#include <string>
#include <vector>

namespace ddl {

class StoreServer;

class StoreClient {
public:
    StoreClient() = default;
    ~StoreClient();
    // connect with retries until `timeout_ms` elapses; returns false on failure
    bool connect_to(const std::string& host, int port, int timeout_ms);
    bool set(const std::string& key, const std::string& value);
    // blocks (server side) until the key exists or timeout; returns false on timeout/error
    bool get(const std::string& key, std::string* value, int timeout_ms);
    bool add(const std::string& key, int64_t delta, int64_t* result);
    bool del(const std::string& key);
    void close_fd();
    bool ok() const { return fd_ >= 0; }

private:
    int fd_ = -1;
};

// port 0 = ephemeral; loopback_only binds 127.0.0.1 instead of every interface
StoreServer* store_server_start(int port, int* bound_port, bool loopback_only = false);
void store_server_stop(StoreServer* s);

// socket helpers shared with the data plane
bool send_all(int fd, const void* buf, size_t n);
bool recv_all(int fd, void* buf, size_t n, int timeout_ms = -1);
int tcp_listen(int port, int* bound_port, bool loopback_only = false);
int tcp_connect(const std::string& host, int port, int timeout_ms);
int tcp_accept(int listen_fd, int timeout_ms);

}  // namespace ddl
