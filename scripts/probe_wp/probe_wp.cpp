// Probe (round 4): can an ordinary user track writes to a pinned-in-place host range with userfaultfd write-protect in
// asynchronous mode + the PAGEMAP_SCAN ioctl (Linux >= 6.7)?  Prints what works.  Build: hipcc probe_wp.cpp -o probe_wp
#include <hip/hip_runtime.h>
#include <cerrno>
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <fcntl.h>
#include <sys/ioctl.h>
#include <sys/mman.h>
#include <sys/syscall.h>
#include <unistd.h>
#include <chrono>
#include <linux/userfaultfd.h>

#ifndef UFFD_USER_MODE_ONLY
#define UFFD_USER_MODE_ONLY 1
#endif
#ifndef UFFD_FEATURE_WP_UNPOPULATED
#define UFFD_FEATURE_WP_UNPOPULATED (1 << 13)
#endif
#ifndef UFFD_FEATURE_WP_ASYNC
#define UFFD_FEATURE_WP_ASYNC (1 << 15)
#endif
// include/uapi/linux/fs.h (6.7+)
struct pm_scan_arg_ { uint64_t size, flags, start, end, walk_end, vec, vec_len, max_pages, category_inverted, category_mask, category_anyof_mask, return_mask; };
struct page_region_ { uint64_t start, end, categories; };
#define PAGEMAP_SCAN_ _IOWR('f', 16, struct pm_scan_arg_)
#define PAGE_IS_WPALLOWED_ (1 << 0)
#define PAGE_IS_WRITTEN_ (1 << 1)
#define PM_SCAN_WP_MATCHING_ (1 << 0)
#define PM_SCAN_CHECK_WPASYNC_ (1 << 1)

static double now() { return std::chrono::duration<double>(std::chrono::steady_clock::now().time_since_epoch()).count(); }

int main() {
  const size_t n = 512ull << 20;
  int uffd = (int)syscall(SYS_userfaultfd, O_CLOEXEC | O_NONBLOCK | UFFD_USER_MODE_ONLY);
  printf("userfaultfd(USER_MODE_ONLY): fd %d errno %d (%s)\n", uffd, uffd < 0 ? errno : 0, uffd < 0 ? strerror(errno) : "ok");
  if (uffd < 0) return 1;
  struct uffdio_api api; memset(&api, 0, sizeof api);
  api.api = UFFD_API; api.features = UFFD_FEATURE_WP_ASYNC | UFFD_FEATURE_WP_UNPOPULATED;
  int rc = ioctl(uffd, UFFDIO_API, &api);
  printf("UFFDIO_API(WP_ASYNC | WP_UNPOPULATED): rc %d errno %d features 0x%llx\n", rc, rc ? errno : 0, (unsigned long long)api.features);
  if (rc) return 2;
  char* p = (char*)mmap(nullptr, n, PROT_READ | PROT_WRITE, MAP_PRIVATE | MAP_ANONYMOUS, -1, 0);
  memset(p, 1, n);
  int pin = getenv("PROBE_PIN") ? atoi(getenv("PROBE_PIN")) : 1;
  if (pin) { hipError_t e = hipHostRegister(p, n, hipHostRegisterDefault); printf("hipHostRegister: %s\n", hipGetErrorString(e)); }
  struct uffdio_register reg; memset(&reg, 0, sizeof reg);
  reg.range.start = (uint64_t)p; reg.range.len = n; reg.mode = UFFDIO_REGISTER_MODE_WP;
  rc = ioctl(uffd, UFFDIO_REGISTER, &reg);
  printf("UFFDIO_REGISTER(MODE_WP): rc %d errno %d (%s)\n", rc, rc ? errno : 0, rc ? strerror(errno) : "ok");
  if (rc) return 3;
  int pm = open("/proc/self/pagemap", O_RDONLY);
  auto scan = [&](bool arm, uint64_t* written_pages) -> int {
    static page_region_ vec[4096];
    pm_scan_arg_ a; memset(&a, 0, sizeof a);
    a.size = sizeof a; a.flags = (arm ? PM_SCAN_WP_MATCHING_ : 0) | PM_SCAN_CHECK_WPASYNC_;
    a.start = (uint64_t)p; a.end = (uint64_t)p + n; a.vec = (uint64_t)vec; a.vec_len = 4096;
    a.category_mask = arm ? 0 : PAGE_IS_WRITTEN_; a.category_anyof_mask = 0; a.return_mask = PAGE_IS_WRITTEN_;
    if (arm) { a.category_mask = 0; a.category_anyof_mask = PAGE_IS_WRITTEN_; }
    long r = ioctl(pm, PAGEMAP_SCAN_, &a);
    uint64_t w = 0;
    for (long i = 0; i < r; ++i) w += (vec[i].end - vec[i].start) / 4096;
    if (written_pages) *written_pages = w;
    return (int)r;
  };
  uint64_t w = 0;
  double t0 = now();
  int r = scan(true, &w);
  printf("PAGEMAP_SCAN arm (WP written pages): regions %d errno %d written-before %llu pages, %.3f ms\n", r, r < 0 ? errno : 0, (unsigned long long)w, (now() - t0) * 1e3);
  if (r < 0) return 4;
  t0 = now(); r = scan(false, &w);
  printf("scan after arm, no write: regions %d written %llu pages, %.3f ms\n", r, (unsigned long long)w, (now() - t0) * 1e3);
  p[12345678] = 7; p[n - 1] = 9;
  t0 = now(); r = scan(false, &w);
  printf("scan after 2 CPU writes: regions %d written %llu pages, %.3f ms\n", r, (unsigned long long)w, (now() - t0) * 1e3);
  // a write by the kernel on behalf of the process (read() into the range)
  r = scan(true, &w);
  int zf = open("/dev/zero", O_RDONLY);
  ssize_t got = read(zf, p + (100 << 20), 8192);
  r = scan(false, &w);
  printf("scan after read() of %zd bytes into the range: regions %d written %llu pages (errno of read %d)\n", got, r, (unsigned long long)w, got < 0 ? errno : 0);
  // a DMA write (device -> host) into the armed range, and a DMA read from it
  if (pin) {
    void* d = nullptr; (void)hipMalloc(&d, 64 << 20);
    r = scan(true, &w);
    hipError_t e1 = hipMemcpy(d, p, 64 << 20, hipMemcpyHostToDevice);
    r = scan(false, &w);
    printf("after H2D DMA from the armed range (%s): written %llu pages\n", hipGetErrorString(e1), (unsigned long long)w);
    hipError_t e2 = hipMemcpy(p + (200 << 20), d, 64 << 20, hipMemcpyDeviceToHost);
    r = scan(false, &w);
    printf("after D2H DMA into the armed range (%s): written %llu pages; first byte there %d\n", hipGetErrorString(e2), (unsigned long long)w, p[200 << 20]);
    t0 = now();
    for (int k = 0; k < 10; ++k) { r = scan(true, &w); }
    printf("arm + scan of 512 MB: %.3f ms per call\n", (now() - t0) * 1e2);
    // write throughput into an armed range (first touch of every page faults once)
    r = scan(true, &w);
    t0 = now(); memset(p, 3, n); double t1 = now() - t0;
    t0 = now(); memset(p, 4, n); double t2 = now() - t0;
    printf("memset of the armed range: %.1f ms (then %.1f ms unarmed)\n", t1 * 1e3, t2 * 1e3);
  }
  return 0;
}
