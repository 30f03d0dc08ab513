// Verifies that div_exact() of rt355.hip (q = a*rb; q + fma(-q,b,a)*rb with rb = RN(1/b)) equals the
// IEEE quotient a/b bit for bit: exhaustively for x + 0.5 over every width <= 16384, and on 1.9e9 random
// keyed-RNG style numerators.  gcc -O2 -mfma -ffp-contract=off -o check_div check_div.c -lm && ./check_div
#include <math.h>
#include <stdio.h>
#include <stdint.h>
#include <stdlib.h>
static inline float div3(float a, float b, float rb){ float q=a*rb; float r=fmaf(-q,b,a); return fmaf(r,rb,q);}
int main(){
  long bad=0, tot=0;
  // exhaustive: mode C, every width 1..16384, every x
  for(int W=1; W<=16384; ++W){ float fw=(float)W; float rb=(float)(1.0/(double)W);
    for(int x=0;x<W;++x){ float a=(float)x+0.5f; float t=a/fw; float d=div3(a,fw,rb); tot++; if(t!=d){ if(bad<10) printf("C W=%d x=%d %a %a\n",W,x,t,d); bad++; } } }
  printf("mode C: %ld bad of %ld\n", bad, tot);
  // random: x + draw (24-bit draws) for assorted widths
  uint64_t s=88172645463325252ull; bad=0; tot=0;
  int Ws[]={640,480,1920,1080,3840,2160,7680,4320,1000,333,17,4097,8191,16383,1,2,3,5,7};
  for(int k=0;k<19;++k){ int W=Ws[k]; float fw=(float)W; float rb=(float)(1.0/(double)W);
    for(long i=0;i<100000000L;++i){ s^=s<<13; s^=s>>7; s^=s<<17; int x=(int)((s>>40)%W); float dr=(float)((s>>8)&0xFFFFFF)*(1.0f/16777216.0f);
      float a=(float)x+dr; float t=a/fw; float d=div3(a,fw,rb); tot++; if(t!=d){ if(bad<10) printf("K W=%d a=%a %a %a\n",W,a,t,d); bad++; } } }
  printf("mode K: %ld bad of %ld\n", bad, tot);
  return 0; }
