"""Block-level traffic model of the gather-scatter k_gs<3> on an 8x8x8-element box, lx1 = 8: which thread block fetches which 128-byte line, for the
natural, slab-permuted (xp), pair-interleaved (xp2 = NLG_XP_LAYOUT=1) and face-grouped layouts (DESIGN.md section 5, round-4 lessons).  CPU only."""
import numpy as np, sys
import os; sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from neklab_amd.mesh import box_mesh
N=8
hm = box_mesh((8,8,8), N)
E=hm.glo_num.shape[0]; g=hm.glo_num.reshape(E,-1)
a,j,k=np.meshgrid(np.arange(N),np.arange(N),np.arange(N),indexing='ij')
a=a.ravel(); j=j.ravel(); k=k.ravel(); nat=a+N*j+N*N*k
def sp_tab(N):
    order=[]; used=np.zeros(N*N,bool)
    def put(a,j): order.append(a+N*j); used[a+N*j]=True
    for jj in range(N): put(0,jj)
    for jj in range(N): put(N-1,jj)
    def filler(c):
        for jj in range(1,N-1):
            for aa in range(1,N-1):
                if c>0 and not used[aa+N*jj]: put(aa,jj); c-=1
    for aa in range(1,N-1): put(aa,0)
    filler(2)
    for aa in range(1,N-1): put(aa,N-1)
    filler(2)
    filler(N*N)
    tab=np.zeros(N*N,int)
    for q,o in enumerate(order): tab[o]=q
    return tab
tab=sp_tab(N)
def slot_xp(a,j,k): return N*N*k+tab[a+N*j]
def slot_xp2(a,j,k):
    # slabs 1..6 interleaved in pairs (1,2),(3,4),(5,6): within a 128-double block, each 8-double row of slab k sits next to the same row of its partner
    s=np.empty_like(a)
    t=tab[a+N*j]              # position within slab: row r = t//8, col c = t%8
    r=t//8; c=t%8
    inter=(k>=1)&(k<=6)
    pairidx=(k-1)//2; half=(k-1)%2
    s[inter]=64+pairidx[inter]*128 + r[inter]*16 + half[inter]*8 + c[inter]
    # slabs 0 and 7: face-interior (36) first, then edges (24), corners (4)
    for kk,base in ((0,0),(7,64+3*128)):
        m=(k==kk)
        aa=a[m]; jj=j[m]
        inte=(aa>0)&(aa<N-1)&(jj>0)&(jj<N-1)
        cor=((aa==0)|(aa==N-1))&((jj==0)|(jj==N-1))
        edge=~inte&~cor
        loc=np.empty(m.sum(),int)
        loc[inte]=np.arange(inte.sum()); 
        # edges grouped by which edge: x0,x7 (a fixed, j=1..6), y0,y7
        eid=np.where(aa[edge]==0,0,np.where(aa[edge]==N-1,1,np.where(jj[edge]==0,2,3)))
        pos=np.where(eid<2,jj[edge]-1,aa[edge]-1)
        loc[edge]=36+eid*6+pos
        loc[cor]=60+np.arange(cor.sum())
        s[m]=base+loc
    return s
def slot_fg(a,j,k):
    M=N-2
    ba=(a==0)|(a==N-1); bj=(j==0)|(j==N-1); bk=(k==0)|(k==N-1)
    sa=(a==N-1).astype(int); sj=(j==N-1).astype(int); sk=(k==N-1).astype(int)
    nb=ba.astype(int)+bj+bk
    out=np.zeros_like(a)
    m=nb==3; out[m]=(sa+2*sj+4*sk)[m]
    m=(nb==2)&~ba; out[m]=(8+(0+sj+2*sk)*M+(a-1))[m]
    m=(nb==2)&ba&~bj; out[m]=(8+(4+sa+2*sk)*M+(j-1))[m]
    m=(nb==2)&ba&bj; out[m]=(8+(8+sa+2*sj)*M+(k-1))[m]
    fb=8+12*M
    m=(nb==1)&ba; out[m]=(fb+(0+sa)*M*M+(j-1)+M*(k-1))[m]
    m=(nb==1)&~ba&bj; out[m]=(fb+(2+sj)*M*M+(a-1)+M*(k-1))[m]
    m=(nb==1)&~ba&~bj; out[m]=(fb+(4+sk)*M*M+(a-1)+M*(j-1))[m]
    m=nb==0; out[m]=(fb+6*M*M+(a-1)+M*((j-1)+M*(k-1)))[m]
    return out
lab=g.ravel()
order=np.argsort(lab,kind='stable'); ls=lab[order]
u,start,cnt=np.unique(ls,return_index=True,return_counts=True)
val=np.repeat(cnt,cnt); 
elem=order//N**3; loc=order%N**3
# fetcher id: for pairs: (min elem, max elem) -> face id; others: the label itself (+offset)
gid=np.repeat(np.arange(len(u)),cnt)
emin=np.minimum.reduceat(elem,start); emax=np.maximum.reduceat(elem,start)
face=np.repeat(emin*E+emax,cnt)
fetch=np.where(val==2, face, E*E+gid)
sh=val>1
for name,fn in (('natural',lambda a,j,k:a+N*j+N*N*k),('xp',slot_xp),('xp2',slot_xp2),('fg',slot_fg)):
    sl=np.empty(N**3,int); sl[nat]=fn(a,j,k)
    assert len(np.unique(sl))==N**3, name
    pos=elem*N**3+sl[loc]
    lines=pos[sh]//16
    pairs=np.unique(np.stack([lines,fetch[sh]]),axis=1).shape[1]
    print(name,'line-fetches x128B / useful bytes =', round(pairs*128/(8*sh.sum()),3), ' union lines ratio', round(len(np.unique(lines))*128/(8*sh.sum()),3))

print('--- block-level model: fetcher = thread block of 256 (general groups, quads: 1 per thread; pairs: 2 per thread), groups ordered by smallest position')
for name,fn in (('natural',lambda a,j,k:a+N*j+N*N*k),('xp',slot_xp),('xp2',slot_xp2),('fg',slot_fg)):
    sl=np.empty(N**3,int); sl[nat]=fn(a,j,k)
    pos=elem*N**3+sl[loc]
    # per group: min pos, valence
    gmin=np.minimum.reduceat(pos,start)
    v=cnt
    tot=0; useful=0
    blockbase=0
    res={}
    for cls,sel,per in (('rest',(v>1)&(v!=2)&(v!=4),1),('quads',v==4,1),('pairs',v==2,2)):
        ids=np.flatnonzero(sel)
        ids=ids[np.argsort(gmin[ids],kind='stable')]
        thread=np.arange(len(ids))//per
        block=blockbase+thread//256
        blockbase=block.max()+1 if len(ids) else blockbase
        # expand to copies
        gb=np.empty(len(u),int); gb[:]=-1; gb[ids]=block
        cb=np.repeat(gb,cnt)
        m=cb>=0
        lines=pos[m]//16
        pairs_=np.unique(np.stack([lines,cb[m]]),axis=1).shape[1]
        res[cls]=(pairs_*128, 8*m.sum())
        tot+=pairs_*128; useful+=8*m.sum()
    print(name, {k_:round(v_[0]/max(v_[1],1),2) for k_,v_ in res.items()}, 'total', round(tot/useful,3))
