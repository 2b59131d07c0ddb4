// hc-edgecalc — the launcher.  The program itself (the reference binary's flag surface, the stage, the output files) is hc_cli_main in
// libhcedge.so (cli/hc_cli.cpp); this executable links neither that library nor the HIP runtime, so that its resident form starts in a
// millisecond:
//   hc-edgecalc <arguments>                 loads libhcedge.so from its own directory and runs hc_cli_main (a process per call, as before)
//   hc-edgecalc --resident <arguments>      forwards the arguments, the working directory, the HC_* environment and its stdout / stderr
//                                           descriptors to the user's resident process (started on first use, as a fresh child of THIS
//                                           process, which never touches the GPU) and exits with the code that comes back
//   hc-edgecalc --resident_stop             asks the resident process to leave
//   hc-edgecalc --resident_daemon <socket>  the resident process itself (hc_cli_daemon); not for users
// Socket: $HC_RESIDENT_DIR, else $XDG_RUNTIME_DIR/hc-edgecalc, else /tmp/hc-edgecalc-<uid>; idle time-out HC_RESIDENT_IDLE_S (600).
// Clients whose HIP_ / ROCR_ / CUDA_VISIBLE_DEVICES or GPU_DEVICE_ORDINAL differ get resident processes of their own (a sub-directory
// per setting).  A job sees its client's HC_* variables, with one exception: the experiment-only launch knobs (HC_COOP_DMA, HC_WAVE_QUEUE,
// HC_GRID_MULT, HC_COOP_DEPTH, HC_COOP_WG_PER_CU) are read once per process and stay what the first job saw.
// Argv contract: /root/reference/scripts/pipeline_per_stage.py:223-247,272-298 — `--resident` is the one extra word.
#include <dlfcn.h>
#include <errno.h>
#include <fcntl.h>
#include <signal.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <sys/socket.h>
#include <sys/stat.h>
#include <sys/un.h>
#include <sys/wait.h>
#include <time.h>
#include <unistd.h>

#include <string>
#include <vector>

extern char** environ;
static const uint32_t kResidentMagic = 0x48435235u;

static std::string self_path() {
    char buf[4096];
    const ssize_t n = readlink("/proc/self/exe", buf, sizeof buf - 1);
    if (n <= 0) return "hc-edgecalc";
    buf[n] = 0;
    return buf;
}

static void* load_library() {
    std::string p = self_path();
    p = p.substr(0, p.rfind('/') + 1) + "libhcedge.so";
    void* h = dlopen(p.c_str(), RTLD_NOW | RTLD_GLOBAL);
    if (!h) h = dlopen("libhcedge.so", RTLD_NOW | RTLD_GLOBAL);
    if (!h) fprintf(stderr, "hc-edgecalc: cannot load libhcedge.so: %s\n", dlerror());
    return h;
}

// what the library next to this executable looks like right now: a resident process that loaded another one (the library was rebuilt
// under it) must not answer for it
static uint64_t library_stamp() {
    std::string p = self_path();
    p = p.substr(0, p.rfind('/') + 1) + "libhcedge.so";
    struct stat sb;
    if (stat(p.c_str(), &sb) != 0) return 0;
    return (uint64_t)sb.st_mtim.tv_sec * 1000000007ull + (uint64_t)sb.st_mtim.tv_nsec + ((uint64_t)sb.st_size << 20);
}

static std::string socket_dir() {
    if (const char* d = getenv("HC_RESIDENT_DIR")) return d;
    if (const char* d = getenv("XDG_RUNTIME_DIR")) return std::string(d) + "/hc-edgecalc";
    return "/tmp/hc-edgecalc-" + std::to_string((unsigned)getuid());
}

// Which devices a process sees is fixed when its HIP runtime starts: two clients with different *_VISIBLE_DEVICES must never share one
// resident process (round-5 advisor: they landed silently on the first client's device).  The variables that decide it are part of the
// resident process's identity — a sub-directory of the socket directory named after their hash; none set: the directory itself.
static std::string visibility_tag() {
    static const char* const names[] = {"HIP_VISIBLE_DEVICES", "ROCR_VISIBLE_DEVICES", "CUDA_VISIBLE_DEVICES", "GPU_DEVICE_ORDINAL"};
    uint64_t h = 1469598103934665603ull;  // FNV-1a over name=value of the ones that are set
    bool any = false;
    for (const char* n : names) {
        const char* v = getenv(n);
        if (!v) continue;
        any = true;
        for (const char* c = n; *c; c++) h = (h ^ (unsigned char)*c) * 1099511628211ull;
        h = (h ^ (unsigned char)'=') * 1099511628211ull;
        for (const char* c = v; *c; c++) h = (h ^ (unsigned char)*c) * 1099511628211ull;
        h = (h ^ 0xFFu) * 1099511628211ull;
    }
    if (!any) return "";
    char buf[32];
    snprintf(buf, sizeof buf, "vis-%016llx", (unsigned long long)h);
    return buf;
}

// a directory of this user's, mode 0700, and a directory itself — not a symbolic link somebody else put there (lstat)
static bool own_private_dir(const std::string& dir) {
    mkdir(dir.c_str(), 0700);
    struct stat sb;
    return lstat(dir.c_str(), &sb) == 0 && S_ISDIR(sb.st_mode) && sb.st_uid == getuid() && !(sb.st_mode & 077);
}

static bool write_all(int fd, const void* p, size_t n) {
    const char* c = (const char*)p;
    while (n) {
        const ssize_t k = write(fd, c, n);
        if (k <= 0) {
            if (k < 0 && errno == EINTR) continue;
            return false;
        }
        c += k;
        n -= (size_t)k;
    }
    return true;
}

static int connect_to(const std::string& path) {
    const int s = socket(AF_UNIX, SOCK_STREAM | SOCK_CLOEXEC, 0);
    sockaddr_un addr;
    memset(&addr, 0, sizeof addr);
    addr.sun_family = AF_UNIX;
    if (s < 0 || path.size() >= sizeof addr.sun_path) return -1;
    strcpy(addr.sun_path, path.c_str());
    if (connect(s, (sockaddr*)&addr, sizeof addr) != 0) {
        close(s);
        return -1;
    }
    return s;
}

// a fresh child process (double fork: it outlives us and is nobody's zombie) that becomes the resident process
static void spawn_daemon(const std::string& dir, const std::string& sock) {
    const pid_t p = fork();
    if (p < 0) return;
    if (p == 0) {
        if (fork() != 0) _exit(0);
        setsid();
        const int log = open((dir + "/resident.log").c_str(), O_CREAT | O_WRONLY | O_APPEND, 0600);
        const int nul = open("/dev/null", O_RDONLY);
        if (nul >= 0) dup2(nul, 0);
        if (log >= 0) {
            dup2(log, 1);
            dup2(log, 2);
        }
        const std::string exe = self_path();
        execl(exe.c_str(), "hc-edgecalc", "--resident_daemon", sock.c_str(), (char*)nullptr);
        _exit(127);
    }
    int st;
    waitpid(p, &st, 0);
}

static const int32_t kStaleLibrary = -1000;  // the resident process's answer when its library is not the one next to the executable any more

static int client_once(int argc, char** argv, bool stop);

static int client(int argc, char** argv, bool stop) {
    signal(SIGPIPE, SIG_IGN);  // a resident process that has hung up is an answer to read, not a signal to die of
    int code = client_once(argc, argv, stop);
    for (int k = 0; k < 3 && code == kStaleLibrary; k++) {  // it has left: the next call starts one with the library as it is now
        timespec ts{0, 20000000};
        nanosleep(&ts, nullptr);
        code = stop ? 0 : client_once(argc, argv, stop);
    }
    return code == kStaleLibrary ? 1 : code;
}

static int client_once(int argc, char** argv, bool stop) {
    std::string dir = socket_dir();
    const std::string vis = visibility_tag();
    bool ok_dir = own_private_dir(dir);
    if (ok_dir && !vis.empty()) {
        dir += "/" + vis;
        ok_dir = own_private_dir(dir);
    }
    const std::string sock = dir + "/sock";
    if (!ok_dir) {
        fprintf(stderr, "hc-edgecalc --resident: %s must be a directory (not a link) of this user's with mode 0700\n", dir.c_str());
        return 1;
    }
    int s = connect_to(sock);
    if (s < 0 && stop) return 0;  // nobody to stop
    if (s < 0) {
        spawn_daemon(dir, sock);
        for (int k = 0; k < 3000 && s < 0; k++) {  // it binds before it loads anything: a few milliseconds
            timespec ts{0, 2000000};
            nanosleep(&ts, nullptr);
            s = connect_to(sock);
        }
    }
    if (s < 0) {
        fprintf(stderr, "hc-edgecalc --resident: no resident process at %s (see %s/resident.log)\n", sock.c_str(), dir.c_str());
        return 1;
    }
    std::vector<std::string> args, env;
    for (int i = 0; i < argc; i++)
        if (strcmp(argv[i], "--resident") != 0 && strcmp(argv[i], "--resident_stop") != 0) args.push_back(argv[i]);
    for (char** e = environ; *e; e++)
        if (strncmp(*e, "HC_", 3) == 0 && strncmp(*e, "HC_RESIDENT_", 12) != 0) env.push_back(*e);
    char cwd[4096];
    if (!getcwd(cwd, sizeof cwd)) strcpy(cwd, "/");
    const uint64_t stamp = library_stamp();
    uint32_t head[7] = {kResidentMagic, (uint32_t)args.size(), (uint32_t)env.size(), (uint32_t)strlen(cwd), stop ? 1u : 0u, (uint32_t)stamp, (uint32_t)(stamp >> 32)};
    int fds[2] = {1, 2};
    {
        iovec iov{head, sizeof head};
        char ctl[CMSG_SPACE(sizeof fds)];
        memset(ctl, 0, sizeof ctl);
        msghdr mh;
        memset(&mh, 0, sizeof mh);
        mh.msg_iov = &iov;
        mh.msg_iovlen = 1;
        mh.msg_control = ctl;
        mh.msg_controllen = sizeof ctl;
        cmsghdr* cm = CMSG_FIRSTHDR(&mh);
        cm->cmsg_level = SOL_SOCKET;
        cm->cmsg_type = SCM_RIGHTS;
        cm->cmsg_len = CMSG_LEN(sizeof fds);
        memcpy(CMSG_DATA(cm), fds, sizeof fds);
        fflush(stdout);
        fflush(stderr);
        if (sendmsg(s, &mh, 0) != (ssize_t)sizeof head) {
            fprintf(stderr, "hc-edgecalc --resident: cannot talk to the resident process\n");
            return 1;
        }
    }
    auto send_str = [&](const std::string& v) {
        const uint32_t n = (uint32_t)v.size();
        return write_all(s, &n, sizeof n) && write_all(s, v.data(), n);
    };
    if (!stop) {  // (a resident process that answers "stale library" at once stops reading: the answer is read all the same)
        bool ok = send_str(cwd);
        for (const std::string& a : args) ok = ok && send_str(a);
        for (const std::string& e : env) ok = ok && send_str(e);
    }
    int32_t code = 1;
    size_t got = 0;
    while (got < sizeof code) {
        const ssize_t k = read(s, (char*)&code + got, sizeof code - got);
        if (k <= 0) {
            if (k < 0 && errno == EINTR) continue;
            break;
        }
        got += (size_t)k;
    }
    if (got != sizeof code) {
        fprintf(stderr, "hc-edgecalc --resident: the resident process went away before it answered (see %s/resident.log)\n", dir.c_str());
        return 1;
    }
    return code;
}

int main(int argc, char** argv) {
    bool resident = false, stop = false;
    for (int i = 1; i < argc; i++) {
        if (!strcmp(argv[i], "--resident")) resident = true;
        if (!strcmp(argv[i], "--resident_stop")) stop = true;
    }
    if (resident || stop) return client(argc, argv, stop);
    void* lib = load_library();
    if (!lib) return 1;
    if (argc >= 3 && !strcmp(argv[1], "--resident_daemon")) {
        typedef int (*daemon_fn)(const char*, int, unsigned long long);
        daemon_fn fn = (daemon_fn)dlsym(lib, "hc_cli_daemon");
        if (!fn) return 1;
        const char* idle = getenv("HC_RESIDENT_IDLE_S");
        return fn(argv[2], idle ? atoi(idle) : 600, (unsigned long long)library_stamp());
    }
    typedef int (*main_fn)(int, char**, void (*)(int, void*), void*);
    main_fn fn = (main_fn)dlsym(lib, "hc_cli_main");
    if (!fn) {
        fprintf(stderr, "hc-edgecalc: libhcedge.so has no hc_cli_main\n");
        return 1;
    }
    return fn(argc, argv, nullptr, nullptr);
}
