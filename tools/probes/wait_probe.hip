// Developer probe: does hipStreamWaitEvent block the calling thread?  (hipcc --offload-arch=gfx950 -O2 wait_probe.hip -o wait_probe)
#include <hip/hip_runtime.h>
#include <chrono>
#include <cstdio>
__global__ void spin( long long cycles, int* out ) { const long long t0 = wall_clock64(); while( wall_clock64() - t0 < cycles ) {} if( out ) out[0] = 1; }
static double now() { return std::chrono::duration<double, std::milli>( std::chrono::steady_clock::now().time_since_epoch() ).count(); }
int main()
{
  hipStream_t a, b, c; hipStreamCreateWithFlags( &a, hipStreamNonBlocking ); hipStreamCreateWithFlags( &b, hipStreamNonBlocking ); hipStreamCreateWithFlags( &c, hipStreamNonBlocking );
  int* d; hipMalloc( &d, 64 );
  const long long ms5 = 100000 * 5;      // wall_clock64 ticks at 100 MHz
  for( int flagsCase = 0; flagsCase < 2; flagsCase++ )
  for( int busy = 0; busy < 3; busy++ )
  for( int rep = 0; rep < 3; rep++ )
  {
    hipEvent_t e; hipEventCreateWithFlags( &e, flagsCase ? hipEventDisableTiming : hipEventDefault );
    hipDeviceSynchronize();
    double t0 = now();
    hipLaunchKernelGGL( spin, dim3( 1 ), dim3( 64 ), 0, a, ms5, d );
    hipEventRecord( e, a );
    double t1 = now();
    if( busy == 1 ) hipLaunchKernelGGL( spin, dim3( 1 ), dim3( 64 ), 0, b, ms5 / 5, d + 1 );       // b has work in flight
    if( busy == 2 ) { hipLaunchKernelGGL( spin, dim3( 1 ), dim3( 64 ), 0, b, 1000, d + 1 ); hipStreamSynchronize( b ); }   // b used, idle again
    double t2 = now();
    hipStreamWaitEvent( b, e, 0 );
    double t3 = now();
    hipLaunchKernelGGL( spin, dim3( 1 ), dim3( 64 ), 0, b, 1000, d + 2 );
    double t4 = now();
    hipStreamWaitEvent( c, e, 0 );       // a second waiter on the same event
    double t5 = now();
    hipDeviceSynchronize();
    double t6 = now();
    printf( "event %s, waiter %s: launch+record %.3f, wait call %.3f ms, launch after %.3f, second waiter %.3f, total %.3f\n", flagsCase ? "no-timing" : "default", busy == 0 ? "idle" : busy == 1 ? "busy" : "used-idle",
            t1 - t0, t3 - t2, t4 - t3, t5 - t4, t6 - t0 );
    hipEventDestroy( e );
  }
  return 0;
}
