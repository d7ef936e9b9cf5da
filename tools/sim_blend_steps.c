// Step-count simulation of blend-kernel decompositions on the config-3 frame (dev tool, not product).
#include <stdio.h>
#include <stdlib.h>
#include <stdint.h>
#include <math.h>
#include <string.h>
#include <omp.h>
static void* rd(const char* f, size_t* n){FILE*fp=fopen(f,"rb");fseek(fp,0,SEEK_END);size_t s=ftell(fp);fseek(fp,0,SEEK_SET);void*p=malloc(s);if(fread(p,1,s,fp)!=s)exit(1);fclose(fp);*n=s;return p;}
int main(){
  size_t s; const int W=1920,H=1080,NTX=120,NTY=68;
  int32_t* nc=rd("n_contrib.bin",&s); int32_t* vals=rd("values.bin",&s); int32_t* tr=rd("tile_ranges.bin",&s);
  float* m2=rd("means_2d.bin",&s); float* ci=rd("cov_2d_inv.bin",&s); float* opa=rd("opa.bin",&s);
  // accumulators
  double quad_steps=0, quad_live=0, quad_open=0, quad_zero=0;
  double c44_rowsteps=0, c44_wavesteps=0, c44_wavesteps64=0, c44_live=0;
  double h84_wavesteps=0, h84_rowsteps=0;   // two 8x4 halves per quad
  double c82_wavesteps=0;                   // eight 8x1?? skip
  double tile_steps=0;                      // unit = whole 16x16 tile (any open pixel passes)
  double fwd_quad_steps=0, fwd_live=0, fwd_c44_wavesteps=0, fwd_c44_rowsteps=0;
  double pair_tile_hits=0;
  double fuse_pairs=0, fuse_pairs_rect=0;    // consecutive steps of a quad with disjoint footprints (exact masks / bounding rectangles)
  #pragma omp parallel for schedule(dynamic,4) reduction(+:quad_steps,quad_live,quad_open,quad_zero,c44_rowsteps,c44_wavesteps,c44_wavesteps64,c44_live,h84_wavesteps,h84_rowsteps,tile_steps,fwd_quad_steps,fwd_live,fwd_c44_wavesteps,fwd_c44_rowsteps,pair_tile_hits,fuse_pairs,fuse_pairs_rect)
  for(int t=0;t<NTX*NTY;++t){
    int tx=t%NTX, ty=t/NTX; int st=tr[2*t], en=tr[2*t+1]; int len=en-st; if(len<=0) continue;
    int rem[256]; char open[256]; float pxs[256], pys[256];
    for(int p=0;p<256;++p){int px=tx*16+(p&15), py=ty*16+(p>>4); pxs[p]=px+0.5f; pys[p]=py+0.5f; int in=(px<W&&py<H); int n=in?nc[py*W+px]:0; rem[p]=n; open[p]=(n>0);}
    unsigned long long prevmask[4]={0,0,0,0}; int prevrect[4][4]; int prevfree[4]={0,0,0,0}, prevfree_r[4]={0,0,0,0};
    // ---------------- backward: from the end
    int nb=(len+255)/256;
    for(int b=nb-1;b>=0;--b){
      int any=0; for(int p=0;p<256;++p) any|=open[p]; if(!any) break;
      int cellhits[16]; memset(cellhits,0,sizeof cellhits); int halfhits[8]; memset(halfhits,0,sizeof halfhits);
      int cellhits64[4][16]; memset(cellhits64,0,sizeof cellhits64);
      int cnt=len-b*256; if(cnt>256)cnt=256;
      for(int j=cnt-1;j>=0;--j){
        int g=vals[st+b*256+j]; float mx=m2[2*g],my=m2[2*g+1],a=ci[3*g],bb=ci[3*g+1],c=ci[3*g+2],o=opa[g];
        // quick reject by bbox of alpha>=1/255: q <= 2 ln(255 o)
        float tau=2.0f*logf(255.0f*o); if(!(tau>0)) continue;
        int qhit[4]={0,0,0,0}, qlive[4]={0,0,0,0}; int chit[16]; memset(chit,0,sizeof chit); int clive=0; int hhit[8]; memset(hhit,0,sizeof hhit);
        int thit=0; unsigned long long fmask[4]={0,0,0,0}; int frect[4][4]; for(int k=0;k<4;++k){frect[k][0]=99;frect[k][1]=-1;frect[k][2]=99;frect[k][3]=-1;}
        for(int p=0;p<256;++p){ if(!open[p]) continue; float dx=pxs[p]-mx, dy=pys[p]-my; float q=a*dx*dx+2*bb*dx*dy+c*dy*dy; if(q<0||q>tau) continue;
          float al=o*expf(-0.5f*q); if(al>0.99f)al=0.99f; if(al<1.0f/255.0f) continue;
          int x=p&15,y=p>>4; int quad=(y>>3)*2+(x>>3); int cell=(y>>2)*4+(x>>2); int half=quad*2+((y&7)>>2);
          { int lx=x&7, ly=y&7; fmask[quad]|=1ull<<(ly*8+lx); if(lx<frect[quad][0])frect[quad][0]=lx; if(lx>frect[quad][1])frect[quad][1]=lx; if(ly<frect[quad][2])frect[quad][2]=ly; if(ly>frect[quad][3])frect[quad][3]=ly; }
          qhit[quad]=1; chit[cell]=1; hhit[half]=1; thit=1;
          rem[p]--; if(rem[p]<0){open[p]=0;} else {qlive[quad]++; clive++;}
        }
        if(thit){tile_steps+=1;pair_tile_hits+=1;}
        for(int k=0;k<4;++k) if(qhit[k]){
          if(prevfree[k] && !(prevmask[k]&fmask[k])){fuse_pairs+=1; prevfree[k]=0;} else {prevfree[k]=1;}
          int dis = prevfree_r[k] && (frect[k][0]>prevrect[k][1] || frect[k][1]<prevrect[k][0] || frect[k][2]>prevrect[k][3] || frect[k][3]<prevrect[k][2]);
          if(dis){fuse_pairs_rect+=1; prevfree_r[k]=0;} else {prevfree_r[k]=1;}
          prevmask[k]=fmask[k]; for(int e=0;e<4;++e) prevrect[k][e]=frect[k][e];
        }
        for(int k=0;k<4;++k) if(qhit[k]){quad_steps+=1; quad_live+=qlive[k]; if(!qlive[k])quad_zero+=1; int op=0; for(int p=0;p<256;++p){int x=p&15,y=p>>4; if(((y>>3)*2+(x>>3))==k) op+=open[p];} quad_open+=op;}
        for(int k=0;k<16;++k) if(chit[k]){cellhits[k]++; cellhits64[j>>6][k]++; c44_rowsteps+=1;}
        for(int k=0;k<8;++k) if(hhit[k]){halfhits[k]++; h84_rowsteps+=1;}
        c44_live+=clive;
      }
      // wave = 8x8 quad = cells {(2qy,2qx),(2qy,2qx+1),(2qy+1,2qx),(2qy+1,2qx+1)}
      for(int qy=0;qy<2;++qy)for(int qx=0;qx<2;++qx){int mxh=0; for(int cy=0;cy<2;++cy)for(int cx=0;cx<2;++cx){int k=(2*qy+cy)*4+2*qx+cx; if(cellhits[k]>mxh)mxh=cellhits[k];} c44_wavesteps+=mxh;
        for(int sb=0;sb<4;++sb){int m=0; for(int cy=0;cy<2;++cy)for(int cx=0;cx<2;++cx){int k=(2*qy+cy)*4+2*qx+cx; if(cellhits64[sb][k]>m)m=cellhits64[sb][k];} c44_wavesteps64+=m;}
        int q=qy*2+qx; int mh=halfhits[2*q]>halfhits[2*q+1]?halfhits[2*q]:halfhits[2*q+1]; h84_wavesteps+=mh;}
    }
    // ---------------- forward: from the front until T<1/255 (use n_contrib as the count of contributors)
    int fr[256]; char fo[256]; for(int p=0;p<256;++p){int px=tx*16+(p&15), py=ty*16+(p>>4); int in=(px<W&&py<H); fr[p]=in?nc[py*W+px]:0; fo[p]=in;}
    // a pixel is open until it has had n_contrib contributors AND was saturated... approximate: open until count reaches n_contrib if saturated;
    // unsaturated pixels stay open to the end of the list.  All pixels saturate in this scene.
    for(int b=0;b<nb;++b){
      int any=0; for(int p=0;p<256;++p) any|=fo[p]; if(!any) break;
      int cellhits[16]; memset(cellhits,0,sizeof cellhits);
      int cnt=len-b*256; if(cnt>256)cnt=256;
      for(int j=0;j<cnt;++j){
        int g=vals[st+b*256+j]; float mx=m2[2*g],my=m2[2*g+1],a=ci[3*g],bb=ci[3*g+1],c=ci[3*g+2],o=opa[g];
        float tau=2.0f*logf(255.0f*o); if(!(tau>0)) continue;
        int qhit[4]={0,0,0,0}; int chit[16]; memset(chit,0,sizeof chit); int live=0;
        for(int p=0;p<256;++p){ if(!fo[p]) continue; float dx=pxs[p]-mx, dy=pys[p]-my; float q=a*dx*dx+2*bb*dx*dy+c*dy*dy; if(q<0||q>tau) continue;
          float al=o*expf(-0.5f*q); if(al>0.99f)al=0.99f; if(al<1.0f/255.0f) continue;
          int x=p&15,y=p>>4; qhit[(y>>3)*2+(x>>3)]=1; chit[(y>>2)*4+(x>>2)]=1; live++; fr[p]--; if(fr[p]<=0) fo[p]=0; }
        for(int k=0;k<4;++k) if(qhit[k]) fwd_quad_steps+=1;
        for(int k=0;k<16;++k) if(chit[k]){cellhits[k]++; fwd_c44_rowsteps+=1;}
        fwd_live+=live;
      }
      for(int qy=0;qy<2;++qy)for(int qx=0;qx<2;++qx){int mxh=0; for(int cy=0;cy<2;++cy)for(int cx=0;cx<2;++cx){int k=(2*qy+cy)*4+2*qx+cx; if(cellhits[k]>mxh)mxh=cellhits[k];} fwd_c44_wavesteps+=mxh;}
    }
  }
  printf("BACKWARD\n quad(8x8) wave-steps %.4g  live/step %.1f  open/step %.1f  zero-live steps %.3g\n", quad_steps, quad_live/quad_steps, quad_open/quad_steps, quad_zero);
  printf(" 4x4 cells: row-steps %.4g  wave-steps(max of 4 rows per 256-batch) %.4g  (per 64-sub-batch %.4g)  ideal rows/4 %.4g  live/row-step %.1f of 16\n", c44_rowsteps, c44_wavesteps, c44_wavesteps64, c44_rowsteps/4, c44_live/c44_rowsteps);
  printf(" 8x4 halves: row-steps %.4g wave-steps(max of 2) %.4g\n", h84_rowsteps, h84_wavesteps);
  printf(" consecutive steps of a quad with disjoint footprints (could share one step): %.4g by exact masks, %.4g by bounding rectangles, of %.4g steps\n", fuse_pairs, fuse_pairs_rect, quad_steps);
  printf(" (tile,Gaussian) pairs with any contribution: %.4g ; live total %.4g -> perfect 64-lane packing %.4g steps\n", pair_tile_hits, c44_live, c44_live/64);
  printf("FORWARD\n quad wave-steps %.4g live total %.4g live/step %.1f ; 4x4: row-steps %.4g wave-steps %.4g\n", fwd_quad_steps, fwd_live, fwd_live/fwd_quad_steps, fwd_c44_rowsteps, fwd_c44_wavesteps);
  return 0;
}
