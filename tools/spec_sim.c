/*
 * spec_sim.c -- CPU experiment behind DESIGN.md 4.1 "why the recurrence is not speculated": does a Costas loop started
 * from a WRONG state ever become bit-identical to the true one?  (If it did quickly and reliably, a frame's 2048
 * serial steps could be cut into segments run in parallel lanes from guessed states and verified exactly.)
 *
 *   python tools/spec_sim_gen.py 64 0.0 50 /tmp/d.bin     # decimated symbols of 64 frames through the oracle
 *   gcc -O2 -ffp-contract=off -Ioracle tools/spec_sim.c -o /tmp/spec_sim -Loracle -l:libqpsk_oracle.so -Wl,-rpath,$PWD/oracle -lm
 *   /tmp/spec_sim /tmp/d.bin 64 1e-3 1e-5                 # perturbation of phase / freq at the restart point
 *
 * Test infrastructure (uses the oracle's qo_costas_step); results in profiles/r02_speculation_sim.txt.
 */
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <math.h>
#include <stdint.h>
#include "qpsk_oracle.h"
#define N 2048
typedef struct { float p, f; } st_t;
static inline uint32_t bits(float x){ uint32_t u; memcpy(&u,&x,4); return u; }
int main(int argc, char **argv){
    const char *fn = argv[1]; int F = atoi(argv[2]);
    double dp = atof(argv[3]), df = atof(argv[4]);
    float *d = malloc(sizeof(float)*2*N*F);
    FILE *fp = fopen(fn,"rb"); if (fread(d,sizeof(float)*2*N,F,fp)!=(size_t)F) return 1; fclose(fp);
    qo_costas c0; qo_costas_create(&c0, (float)(6.283185307179586/100.0f), -1.0f, 1.0f);
    static st_t tr[N+1];
    int hist[64]={0}; int maxc=0, fails=0, total=0; long sum=0;
    int worst_frame=-1, worst_k0=-1;
    for (int f=0; f<F; f++){
        qo_costas c = c0; c.phase=0; c.freq=0;
        const float *df_ = d + (size_t)2*N*f;
        for (int k=0;k<N;k++){ tr[k].p=c.phase; tr[k].f=c.freq; float zr,zi; qo_costas_step(&c, df_[2*k], df_[2*k+1], &zr,&zi);} 
        tr[N].p=c.phase; tr[N].f=c.freq;
        for (int k0=128; k0<N-64; k0+=61){
            for (int sgn=-1; sgn<=1; sgn+=2){
                qo_costas s = c0; s.phase = (float)(tr[k0].p + sgn*dp); s.freq = (float)(tr[k0].f + sgn*df);
                int k=k0, merged=-1;
                for (; k<N; k++){
                    if (bits(s.phase)==bits(tr[k].p) && bits(s.freq)==bits(tr[k].f)) { merged=k-k0; break; }
                    float zr,zi; qo_costas_step(&s, df_[2*k], df_[2*k+1], &zr,&zi);
                }
                total++;
                if (merged<0){ fails++; if (N-k0 > 700) printf("  no merge: frame %d k0 %d (%d steps available) final dphase %g dfreq %g\n", f,k0,N-k0, s.phase-tr[N].p, s.freq-tr[N].f); }
                else { sum+=merged; if (merged>maxc){maxc=merged; worst_frame=f; worst_k0=k0;} int b=merged/32; if(b>63)b=63; hist[b]++; }
            }
        }
    }
    printf("%s dp %g df %g: trials %d, no-merge-before-end %d, mean %ld, max %d (frame %d k0 %d)\n", fn, dp, df, total, fails, total-fails? sum/(total-fails):0, maxc, worst_frame, worst_k0);
    printf("hist (bins of 32 steps): "); for (int b=0;b<64;b++) if (hist[b]) printf("[%d]=%d ", b*32, hist[b]); printf("\n");
    return 0;
}
