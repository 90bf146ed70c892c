"""developer helper: where the host stage of vvr_submit spends its time on THIS machine - a sampled profile (SIGPROF, the interrupted program counter, addr2line)
of 1000 submissions of 4K B pictures of bench.py's mix against the stand-in runtime of tests/hoststub (no GPU; the stand-in's H2D is a memcpy, its kernels are empty),
with bench.py's default tool flags (edge parameters and affine sub-block MVs derived on the device).  Prints ms per picture, then the share of samples per function
and per source line.  Usage: python tools/host_stage_profile.py > profiles/roundN_host_stage_profile.txt"""
import ctypes as C, os, subprocess, sys, time, collections
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
PCPROF_C = r'''
#define _GNU_SOURCE
#include <signal.h>
#include <sys/time.h>
#include <ucontext.h>
#include <stdio.h>
#include <stdlib.h>
#include <stdint.h>
#include <string.h>
#include <dlfcn.h>
static uintptr_t* g_pcs; static volatile size_t g_n; static size_t g_cap;
static void onprof( int sig, siginfo_t* si, void* uc_ ) { ucontext_t* uc = (ucontext_t*) uc_; if( g_n < g_cap ) g_pcs[g_n++] = (uintptr_t) uc->uc_mcontext.gregs[REG_RIP]; }
void pcprof_start( int us ) { g_cap = 1 << 22; g_pcs = malloc( sizeof( uintptr_t ) * g_cap ); g_n = 0;
  struct sigaction sa; memset( &sa, 0, sizeof( sa ) ); sa.sa_sigaction = onprof; sa.sa_flags = SA_SIGINFO | SA_RESTART; sigaction( SIGPROF, &sa, 0 );
  struct itimerval tv = { { 0, us }, { 0, us } }; setitimer( ITIMER_PROF, &tv, 0 ); }
void pcprof_stop( const char* path, void* anySymbolOfLib ) { struct itimerval tv = { { 0, 0 }, { 0, 0 } }; setitimer( ITIMER_PROF, &tv, 0 );
  Dl_info di; dladdr( anySymbolOfLib, &di ); FILE* f = fopen( path, "w" ); size_t in = 0;
  for( size_t i = 0; i < g_n; i++ ) { Dl_info d2; if( dladdr( (void*) g_pcs[i], &d2 ) && d2.dli_fbase == di.dli_fbase ) { fprintf( f, "%lx\n", (unsigned long) ( g_pcs[i] - (uintptr_t) di.dli_fbase ) ); in++; } }
  fclose( f ); fprintf( stderr, "pcprof: %zu samples, %zu in %s\n", (size_t) g_n, in, di.dli_fname ); }
'''
open("/tmp/pcprof.c", "w").write(PCPROF_C)
subprocess.check_call(["gcc", "-O2", "-fPIC", "-shared", "/tmp/pcprof.c", "-o", "/tmp/libpcprof.so", "-ldl"])
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
from vvdec_amd import abi, synth, stream
import bench
import test_host_glue as T
lib = "/tmp/vvr_hoststub_prof.so"
subprocess.check_call(["g++", "-std=c++17", "-O3", "-g", "-fno-omit-frame-pointer", "-fPIC", "-shared", "-pthread", "-Wl,-Bsymbolic", "-I" + T.HIP_INC, "-D__HIP_PLATFORM_AMD__", "-DVVT_NO_LF_STANDIN", "-w", T.SRC, "-o", lib])
L = C.CDLL(lib); P = C.CDLL("/tmp/libpcprof.so")
L.vvr_submit.argtypes = [C.c_void_p, C.c_void_p]; L.vvr_sync.argtypes = [C.c_void_p]
W, H = 3840, 2160
plans, nslots = stream.ra_plan(17, gop=16, seed_poc0_is_external=False, pool=24)
tools = bench._tools(abi) | abi.TOOL_LFP_ON_DEVICE | abi.TOOL_AFFINE_MV_ON_DEVICE
descs = [synth.picture_for_plan(pl, W, H, seed=1234, tool_flags=tools, **bench.MIX) for pl in plans[1:5]]
pics = [d.c() for d in descs]
cfg = abi.Config(); cfg.abi_version = abi.VVR_ABI_VERSION; cfg.max_width = W; cfg.max_height = H; cfg.chroma_format = 1; cfg.bit_depth = 10; cfg.log2_ctu = 7
cfg.num_slots = 24; cfg.num_streams = 8; cfg.host_threads = 0
ctx = C.c_void_p(); assert L.vvr_create(C.byref(cfg), C.byref(ctx)) == 0
for p in pics: L.vvr_submit(ctx, C.byref(p))
L.vvr_sync(ctx)
P.pcprof_start(200)
t0 = time.perf_counter()
for it in range(250):
    for p in pics: assert L.vvr_submit(ctx, C.byref(p)) >= 0
L.vvr_sync(ctx)
dt = time.perf_counter() - t0
P.pcprof_stop.argtypes = [C.c_char_p, C.c_void_p]
P.pcprof_stop(b"/tmp/pcs.txt", C.cast(L.vvr_submit, C.c_void_p))
print("%.2f ms per picture" % (dt / 1000 * 1e3))
pcs = [l.strip() for l in open("/tmp/pcs.txt")]
out = subprocess.run(["addr2line", "-e", lib, "-f", "-C"] + ["0x" + a for a in pcs], capture_output=True, text=True).stdout.split("\n")
funcs = collections.Counter(); lines = collections.Counter()
for i in range(0, len(out) - 1, 2):
    funcs[out[i][:90]] += 1; lines[out[i + 1].split("/")[-1].split(" ")[0]] += 1
n = len(pcs)
print("--- functions"); [print("%5.1f%%  %s" % (100.0 * c / n, f)) for f, c in funcs.most_common(14)]
print("--- lines"); [print("%5.1f%%  %s" % (100.0 * c / n, f)) for f, c in lines.most_common(45)]
