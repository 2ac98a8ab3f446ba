// cavif_mi -- the cavif command line (src/main.rs) on top of libmi_avif.so: same flags, path rules and report line;
// the rayon fan-out over files (src/main.rs:223) becomes mi_ravif_encode_batch (one host thread per MI355X).
//   cavif_mi [-Q n] [-s n] [-j n] [-f] [-o path] [-q] [--dirty-alpha] [--color ycbcr|rgb] [--depth 8|10|auto] IMAGES...
// Differences, deliberate: PNG input only (the reference also reads JPEG through load_image), `--devices a,b,..`
// selects HIP devices (default: all), and there is no CPU fallback -- without a GPU every file fails loudly.
#include <sched.h>
#include <sys/stat.h>
#include <cerrno>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <algorithm>
#include <atomic>
#include <condition_variable>
#include <mutex>
#include <string>
#include <thread>
#include <vector>
#include "../../include/mi_avif.h"

namespace {

struct Input { bool is_stdio = false; std::string path; };
bool exists(const std::string &p) { struct stat st; return stat(p.c_str(), &st) == 0; }
bool is_dir(const std::string &p) { struct stat st; return stat(p.c_str(), &st) == 0 && S_ISDIR(st.st_mode); }
void mkdirs(const std::string &p) {
  for (size_t i = 1; i <= p.size(); i++) if (i == p.size() || p[i] == '/') { const std::string s = p.substr(0, i); mkdir(s.c_str(), 0777); }
}
std::string file_name(const std::string &p) { const size_t s = p.find_last_of('/'); return s == std::string::npos ? p : p.substr(s + 1); }
// Path::extension(): text after the last '.' of the file name, none for names that start with their only dot
bool extension(const std::string &p, std::string *ext) {
  const std::string f = file_name(p); const size_t d = f.find_last_of('.');
  if (d == std::string::npos || d == 0) return false;
  *ext = f.substr(d + 1); return true;
}
std::string with_extension_avif(const std::string &p) {
  std::string e; if (!extension(p, &e)) return p + ".avif";
  return p.substr(0, p.size() - e.size()) + "avif";
}
bool read_all(FILE *f, std::vector<uint8_t> &out) {
  uint8_t buf[1 << 16]; size_t n;
  while ((n = fread(buf, 1, sizeof(buf), f)) > 0) out.insert(out.end(), buf, buf + n);
  return !ferror(f);
}
// rayon::current_num_threads() of a default global pool: RAYON_NUM_THREADS when set, else std::thread::available_parallelism() -- the CPUs this process may run on (its
// affinity mask) bounded by the cgroup CPU quota (v2: cpu.max along the process's cgroup path; v1: cfs quota / period), integer division, at least one
static long cgroup_cpu_quota() {
  long quota = -1;
  std::string path;
  if (FILE *f = fopen("/proc/self/cgroup", "r")) {
    char ln[4096];
    while (fgets(ln, sizeof(ln), f)) if (!strncmp(ln, "0::", 3)) { path = ln + 3; while (!path.empty() && (path.back() == '\n' || path.back() == '\r')) path.pop_back(); }
    fclose(f);
  }
  std::string d = "/sys/fs/cgroup" + (path.empty() || path[0] != '/' ? "/" + path : path);
  while (d.size() > 1 && d.back() == '/') d.pop_back();
  for (;;) {
    if (FILE *f = fopen((d + "/cpu.max").c_str(), "r")) {
      char lim[64]; long per = 0;
      if (fscanf(f, "%63s %ld", lim, &per) == 2 && strcmp(lim, "max") != 0 && per > 0) { const long q = atol(lim) / per; if (quota < 0 || q < quota) quota = q; }
      fclose(f);
    }
    if (d == "/sys/fs/cgroup" || d.size() <= strlen("/sys/fs/cgroup")) break;
    d = d.substr(0, d.find_last_of('/'));
  }
  if (quota < 0) {
    long q = -1, per = 0;
    if (FILE *f = fopen("/sys/fs/cgroup/cpu/cpu.cfs_quota_us", "r")) { if (fscanf(f, "%ld", &q) != 1) q = -1; fclose(f); }
    if (FILE *f = fopen("/sys/fs/cgroup/cpu/cpu.cfs_period_us", "r")) { if (fscanf(f, "%ld", &per) != 1) per = 0; fclose(f); }
    if (q > 0 && per > 0) quota = q / per;
  }
  return quota;
}
int host_threads() {
  if (const char *e = getenv("RAYON_NUM_THREADS")) { const int v = atoi(e); if (v > 0) return v; }
  int n = (int)std::max(1u, std::thread::hardware_concurrency());
  cpu_set_t set; CPU_ZERO(&set);
  if (sched_getaffinity(0, sizeof(set), &set) == 0 && CPU_COUNT(&set) > 0) n = CPU_COUNT(&set);
  const long q = cgroup_cpu_quota();
  if (q >= 0) n = (int)std::min<long>(n, std::max<long>(q, 1));
  return n;
}
int usage(const char *msg) {
  fprintf(stderr, "error: %s\nusage: cavif_mi [-Q quality 1-100] [-s speed 1-10] [-j threads] [-f|--overwrite] [-o path] [-q] [--dirty-alpha]\n"
                  "                [--color ycbcr|rgb] [--depth 8|10|auto] [--devices 0,1,..] [--rdo-passes 1|2] IMAGES...   (\"-\" = stdin/stdout)\n", msg);
  return 1;
}

}  // namespace

#include <chrono>
#include <unistd.h>
#include <malloc.h>
#include <signal.h>
#include <sys/prctl.h>
#include <sys/wait.h>
static double now_s() { return std::chrono::duration<double>(std::chrono::steady_clock::now().time_since_epoch()).count(); }
int main(int argc, char **argv) {
  const double t_start = now_s(); const bool timing = getenv("CAVIF_MI_TIMING") != nullptr;
  // The loaders' buffers (file, inflate output, RGBA: ~8 MB each) come from the heap arenas and stay there: as anonymous mappings every one of them is an
  // mmap + page faults + munmap (a TLB shootdown across the loader threads), all under the address-space lock the HIP runtime needs while it starts.
  mallopt(M_MMAP_THRESHOLD, 32 << 20); mallopt(M_TRIM_THRESHOLD, 1 << 30);
  // CAVIF_MI_BACKGROUND_EXIT=1 (opt-in): device teardown off the caller's clock.  The work runs in a child; once every output is on disk the child reports its
  // status through a pipe, the parent returns it at once, and the child's exit (the kernel unpins the staging, frees the arenas and destroys the queues: ~0.35 s)
  // goes on unattended.  The GPU stays busy with that teardown for a moment after the command has returned: a caller that chains invocations back to back gains
  // nothing, and the orphaned child is reaped by PID 1 (a container whose PID 1 does not reap collects zombies).  Default: one process, teardown included.
  int done_fd = -1;
  if (getenv("CAVIF_MI_BACKGROUND_EXIT") && !getenv("CAVIF_MI_FOREGROUND_EXIT")) {
    int pfd[2];
    if (pipe(pfd) == 0) {
      const pid_t parent = getpid();
      const pid_t child = fork();
      if (child > 0) {
        close(pfd[1]);
        unsigned char st = 0; ssize_t n;
        do n = read(pfd[0], &st, 1); while (n < 0 && errno == EINTR);
        if (n == 1) _exit(st);
        int ws = 0; while (waitpid(child, &ws, 0) < 0 && errno == EINTR) {}                     // the child ended without reporting (usage error, crash): its status is ours
        _exit(WIFEXITED(ws) ? WEXITSTATUS(ws) : 128 + WTERMSIG(ws));
      }
      if (child == 0) {
        close(pfd[0]); done_fd = pfd[1];
        prctl(PR_SET_PDEATHSIG, SIGKILL);
        if (getppid() != parent) _exit(1);                                                     // the parent died between fork and prctl: nobody is waiting
      }
      else { close(pfd[0]); close(pfd[1]); }                                                  // no fork: everything in this process
    }
  }
  float quality = 80.f; int speed = 4, threads = 0, depth = 0, color_model = 0, rdo_passes = 1;
  bool overwrite = false, quiet = false, dirty_alpha = false, have_output = false, output_stdio = false;
  std::string output; std::vector<std::string> images; std::vector<int> devices;
  // clap syntax (src/main.rs:45-110): --name value, --name=value, -n value, -nvalue, -n=value, combined short flags (-fq)
  std::vector<std::string> args;
  bool only_positional = false;
  for (int i = 1; i < argc; i++) {
    const std::string a = argv[i];
    if (only_positional || a == "-" || a.size() < 2 || a[0] != '-') { args.push_back(a); continue; }
    if (a == "--") { only_positional = true; continue; }
    if (a[1] == '-') { const size_t eq = a.find('='); if (eq == std::string::npos) args.push_back(a); else { args.push_back(a.substr(0, eq)); args.push_back(a.substr(eq + 1)); } continue; }
    for (size_t k = 1; k < a.size(); k++) {
      const char c = a[k];
      args.push_back(std::string("-") + c);
      if (c == 'Q' || c == 's' || c == 'j' || c == 'o') {          // takes a value: the rest of the token, if any
        if (k + 1 < a.size()) args.push_back(a.substr(k + 1 + (a[k + 1] == '=' ? 1 : 0)));
        break;
      }
    }
  }
  const int nargs = (int)args.size();
  for (int i = 0; i < nargs; i++) {
    const std::string a = args[i];
    auto value = [&](const char *name) -> const char * { if (i + 1 >= nargs) { usage((std::string("missing value for ") + name).c_str()); exit(1); } return args[++i].c_str(); };
    if (a == "-Q" || a == "--quality") {
      char *end; quality = strtof(value("--quality"), &end);
      if (*end || !(quality >= 1.f && quality <= 100.f)) return usage("quality must be a number between 1 and 100");      // parse_quality, src/main.rs:24-33
    } else if (a == "-s" || a == "--speed") {
      char *end; const long v = strtol(value("--speed"), &end, 10);
      if (*end || v < 1 || v > 10) return usage("speed must be a number between 1 and 10");                                // parse_speed, :35-43
      speed = (int)v;
    } else if (a == "-j" || a == "--threads") { threads = atoi(value("--threads")); if (threads < 0 || threads > 255) return usage("bad thread count"); }
    else if (a == "-f" || a == "--overwrite" || a == "--force") overwrite = true;
    else if (a == "-o" || a == "--output") { output = value("--output"); have_output = true; output_stdio = output == "-"; }
    else if (a == "-q" || a == "--quiet") quiet = true;
    else if (a == "--dirty-alpha") dirty_alpha = true;
    else if (a == "--color") { const std::string v = value("--color"); if (v == "ycbcr") color_model = 0; else if (v == "rgb") color_model = 1; else return usage("bad color type"); }
    else if (a == "--rdo-passes") { rdo_passes = atoi(value("--rdo-passes")); if (rdo_passes < 1 || rdo_passes > 2) return usage("bad --rdo-passes (1 or 2)"); }   // extension: not a cavif flag
    else if (a == "--depth") { const std::string v = value("--depth"); depth = v == "8" ? 8 : v == "10" ? 10 : 0; if (v != "8" && v != "10" && v != "auto") return usage("bad depth"); }
    else if (a == "--devices") { const char *v = value("--devices"); for (const char *p = v; *p;) { devices.push_back((int)strtol(p, (char **)&p, 10)); if (*p == ',') p++; } }
    else if (a.size() > 1 && a[0] == '-' ) return usage(("unknown option " + a).c_str());
    else images.push_back(a);
  }
  if (images.empty()) return usage("Please specify image paths to convert");
  // file filter, src/main.rs:137-160
  std::vector<Input> files;
  for (const std::string &p : images) {
    if (p == "-") { Input in; in.is_stdio = true; files.push_back(in); continue; }
    if (quiet) { char *end; const long v = strtol(p.c_str(), &end, 10); if (!*end && v >= 0 && v <= 255 && !exists(p)) fprintf(stderr, "warning: -q is not for quality, so '%s' is misinterpreted as a file. Use -Q %s\n", p.c_str(), p.c_str()); }
    std::string ext;
    if (extension(p, &ext) && ext == "avif") {
      if (!quiet) {
        if (exists(p)) fprintf(stderr, "warning: ignoring %s, because it's already an AVIF\n", p.c_str());
        else { fprintf(stderr, "warning: Did you mean to use -o %s?\n", p.c_str()); Input in; in.path = p; files.push_back(in); }
      }
      continue;
    }
    Input in; in.path = p; files.push_back(in);
  }
  if (files.empty()) { fprintf(stderr, "error: No PNG/JPEG files specified\n"); return 1; }
  bool use_dir = false;
  if (have_output && !output_stdio) { if (files.size() > 1) mkdirs(output); use_dir = files.size() > 1 || is_dir(output); }

  mi_ravif_encoder enc; mi_ravif_encoder_default(&enc);
  enc.quality = quality;
  enc.alpha_quality = std::fmin((quality + 100.f) / 2.f, quality + quality / 4.f + 2.f);                                   // :115
  enc.speed = (uint8_t)speed; enc.depth = (uint8_t)depth; enc.color_model = (uint8_t)color_model; enc.rdo_passes = rdo_passes;
  enc.alpha_mode = dirty_alpha ? 0 : 1;
  // -j absent or 0: the reference resolves `threads: None` to rayon::current_num_threads() = the host's logical cores (ravif/src/av1encoder.rs:665-668), which bounds
  // the tile target min(T, w*h / min_tile_size^2): the same file on the same host gets the same tiles from `cavif` and from `cavif_mi`
  enc.threads = threads > 0 ? threads : host_threads();

  // load + decide output paths (process(), :169-200); failures are collected per file and reported at the end
  struct Job { std::string in_name, out_path; bool out_stdio = false; uint8_t *rgba = nullptr; uint32_t w = 0, h = 0; std::string error; };
  std::vector<Job> jobs(files.size());
  // the reference loads inside files.into_par_iter() (src/main.rs:223): file reads + PNG decodes fan out over the host cores
  auto load = [&](size_t i) {
    Job &j = jobs[i]; const Input &in = files[i];
    j.in_name = in.is_stdio ? "stdin" : in.path;
    std::vector<uint8_t> data;
    if (in.is_stdio) { if (!read_all(stdin, data)) j.error = "Unable to read stdin"; }
    else { FILE *f = fopen(in.path.c_str(), "rb"); if (!f || !read_all(f, data)) j.error = "Unable to read input image " + in.path + ": " + strerror(errno); if (f) fclose(f); }
    if (j.error.empty()) {
      const int st = mi_png_decode_rgba(data.data(), data.size(), &j.rgba, &j.w, &j.h);
      if (st) j.error = st == MI_UNSUPPORTED ? "unsupported image format (this build reads PNG)" : "corrupt PNG data";
    }
    if (!have_output) { if (in.is_stdio) j.out_stdio = true; else j.out_path = with_extension_avif(in.path); }
    else if (output_stdio) j.out_stdio = true;
    else if (in.is_stdio) j.out_path = output;
    else j.out_path = use_dir ? output + "/" + with_extension_avif(file_name(in.path)) : output;
    if (j.error.empty() && !j.out_stdio && !overwrite && exists(j.out_path)) j.error = j.out_path + " already exists; skipping";
  };
  // loaders (host cores) and the encoder (GPUs) run concurrently: the encoder pulls image i through fetch(), which waits for
  // loader i -- the reference gets the same overlap from rayon's work stealing over process() (src/main.rs:179-223).  Loaders
  // stay at most `window` images ahead of the images the encoder has released (pixels copied to its staging and freed here), so
  // host memory is bounded for any number of files.
  std::mutex mu; std::condition_variable cv; std::vector<char> loaded(files.size(), 0);
  // A 1080p PNG decodes in ~25 ms and one GPU takes an image every ~4 ms: a dozen loaders per GPU keep up with room to spare, and 24 hand the first run
  // over in one round.  More of them only contend with the HIP runtime's start (64 loaders: runtime + first batch object up after 0.55 s instead of 0.2 s,
  // profiles/r05zk_e2e_knobs.txt).  The pool starts before the device count is known and grows once it is.
  const unsigned loaders_per_device = 12, loaders_first = getenv("CAVIF_MI_LOADERS") ? (unsigned)atoi(getenv("CAVIF_MI_LOADERS")) : 2 * loaders_per_device;
  const unsigned loaders_cap = std::max(1u, threads > 0 ? std::min((unsigned)threads, std::thread::hardware_concurrency()) : std::thread::hardware_concurrency());
  const size_t nw = std::min<size_t>(files.size(), std::max(1u, std::min(loaders_first, loaders_cap)));
  std::atomic<size_t> window{ 4 * 32 + nw };                        // widened below once the device count is known (loaders start first)
  size_t released = 0;                                              // guarded by mu: images the encoder is done reading (or that failed to load)
  std::atomic<size_t> next{ 0 };
  std::vector<std::thread> pool;
  auto loader = [&] {
    for (size_t i; (i = next.fetch_add(1)) < files.size();) {
      { std::unique_lock<std::mutex> lk(mu); cv.wait(lk, [&] { return i < released + window; }); }
      load(i);
      { std::lock_guard<std::mutex> lk(mu); loaded[i] = 1; if (!jobs[i].error.empty()) released++; }
      cv.notify_all();
    }
  };
  for (size_t t = 0; t < nw; t++) pool.emplace_back(loader);
  struct Ctx { std::vector<Job> *jobs; std::mutex *mu; std::condition_variable *cv; std::vector<char> *loaded; size_t *released; } ctx{ &jobs, &mu, &cv, &loaded, &released };
  auto fetch = [](void *user, size_t i, mi_image_desc *d) -> int {
    Ctx *c = (Ctx *)user;
    { std::unique_lock<std::mutex> lk(*c->mu); c->cv->wait(lk, [&] { return (*c->loaded)[i] != 0; }); }
    const Job &j = (*c->jobs)[i];
    if (!j.error.empty()) return MI_INVALID_ARGUMENT;           // reported from the job's own message below
    d->pixels = j.rgba; d->width = j.w; d->height = j.h; d->stride_px = j.w; d->channels = 4;
    return MI_OK;
  };
  auto release = [](void *user, size_t i) {
    Ctx *c = (Ctx *)user;
    Job &j = (*c->jobs)[i];
    mi_free(j.rgba); j.rgba = nullptr;
    { std::lock_guard<std::mutex> lk(*c->mu); (*c->released)++; }
    c->cv->notify_all();
  };
  std::vector<mi_encoded_image> enc_out(jobs.size()); std::vector<int> status(jobs.size(), MI_OK);
  if (timing) fprintf(stderr, "[timing] setup %.3f s (tile target bounded by %d threads)\n", now_s() - t_start, enc.threads);
  const size_t ndev_used = devices.empty() ? (size_t)std::max(1, mi_device_count()) : devices.size();      // brings the HIP runtime up while the loaders run
  const size_t nw_all = std::min<size_t>(files.size(), std::min<size_t>(loaders_cap, getenv("CAVIF_MI_LOADERS") ? nw : loaders_per_device * (ndev_used + 1)));
  { std::lock_guard<std::mutex> lk(mu); window = 4 * 32 * ndev_used + nw_all; }
  for (size_t t = nw; t < nw_all; t++) pool.emplace_back(loader);
  cv.notify_all();
  if (timing) fprintf(stderr, "[timing] HIP runtime up %.3f s\n", now_s() - t_start);
  const int rc_all = mi_ravif_encode_stream(&enc, jobs.size(), fetch, release, &ctx, enc_out.data(), status.data(), devices.empty() ? nullptr : devices.data(), (int)devices.size());
  for (auto &t : pool) t.join();
  if (timing) fprintf(stderr, "[timing] encoded %.3f s\n", now_s() - t_start);
  if (rc_all == MI_NO_DEVICE) for (size_t i = 0; i < jobs.size(); i++) if (jobs[i].error.empty()) status[i] = MI_NO_DEVICE;
  std::vector<size_t> who;
  for (size_t i = 0; i < jobs.size(); i++) if (jobs[i].error.empty()) who.push_back(i);
  for (size_t k = 0; k < who.size(); k++) {
    Job &j = jobs[who[k]]; const mi_encoded_image &im = enc_out[who[k]];
    if (status[who[k]] != MI_OK) {
      static const char *names[] = { "ok", "TooFewPixels", "Unsupported", "EncodingError", "invalid argument", "no HIP device (this encoder has no CPU fallback)" };
      j.error = names[status[who[k]] >= 0 && status[who[k]] <= 5 ? status[who[k]] : 3];
      continue;
    }
    if (j.out_stdio) { if (fwrite(im.avif_file, 1, im.avif_len, stdout) != im.avif_len) j.error = "Unable to write output image: stdout"; }
    else {
      if (!quiet) printf("%s: %zuKB (%zuB color, %zuB alpha, %zuB HEIF)\n", j.out_path.c_str(), (im.avif_len + 999) / 1000, im.color_byte_size, im.alpha_byte_size,
                         im.avif_len - im.color_byte_size - im.alpha_byte_size);                                                                   // :213
      FILE *f = fopen(j.out_path.c_str(), "wb");
      if (!f || fwrite(im.avif_file, 1, im.avif_len, f) != im.avif_len) j.error = "Unable to write output image: " + std::string(strerror(errno));
      if (f) fclose(f);
    }
    mi_free(im.avif_file);
  }
  if (timing) fprintf(stderr, "[timing] written %.3f s\n", now_s() - t_start);
  int failures = 0;
  for (Job &j : jobs) {
    if (j.rgba) mi_free(j.rgba);
    if (!j.error.empty()) { failures++; if (!quiet) fprintf(stderr, "error: %s: error: %s\n", j.in_name.c_str(), j.error.c_str()); }
  }
  // every output is on disk: leave without the HIP runtime's teardown (freeing pinned staging and contexts costs ~0.3 s)
  if (timing) {                                                    // wall-clock stamps for a caller that times process start and exit (tools/e2e_timeline.py)
    const double wall = std::chrono::duration<double>(std::chrono::system_clock::now().time_since_epoch()).count();
    fprintf(stderr, "[timing] main entered at %.6f, leaving at %.6f (unix time)\n", wall - (now_s() - t_start), wall);
  }
  fflush(stdout); fflush(stderr);
  if (done_fd >= 0) { const unsigned char st = failures ? 1 : 0; close(0); close(1); close(2); if (write(done_fd, &st, 1) != 1) {} }
  _exit(failures ? 1 : 0);
}
