import numpy as np, sys
sys.path.insert(0,'/root/repo')
from tactile_gym_amd.urdf_compile import *
A='/root/reference/tactile_gym/assets/robot_assets/'
def raster(tri_cam, W,H,fov,near,far):
    # tri_cam: [T,3,3] in camera coords (x right, y up, z = -forward i.e. GL eye space)
    ys=1/np.tan(np.radians(fov)/2)
    dep=np.ones((H,W),np.float64)
    x=tri_cam[...,0]; y=tri_cam[...,1]; z=tri_cam[...,2]
    wc=-z
    ndx=ys*x/wc; ndy=ys*y/wc
    ndz=((near+far)/(near-far)*z + 2*near*far/(near-far))/wc
    sx=(ndx*0.5+0.5)*W; sy=(0.5-ndy*0.5)*H; sz=ndz*0.5+0.5
    px=np.arange(W)+0.5; py=np.arange(H)+0.5
    PX,PY=np.meshgrid(px,py)
    for t in range(tri_cam.shape[0]):
        if (wc[t]<=near*0.5).any(): continue
        x0,x1,x2=sx[t]; y0,y1,y2=sy[t]
        area=(x1-x0)*(y2-y0)-(x2-x0)*(y1-y0)
        if area==0: continue
        xmin=max(int(np.floor(min(x0,x1,x2))),0); xmax=min(int(np.ceil(max(x0,x1,x2))),W)
        ymin=max(int(np.floor(min(y0,y1,y2))),0); ymax=min(int(np.ceil(max(y0,y1,y2))),H)
        if xmin>=xmax or ymin>=ymax: continue
        X=PX[ymin:ymax,xmin:xmax]; Y=PY[ymin:ymax,xmin:xmax]
        w0=((x1-X)*(y2-Y)-(x2-X)*(y1-Y))/area
        w1=((x2-X)*(y0-Y)-(x0-X)*(y2-Y))/area
        w2=1-w0-w1
        inside=(w0>=0)&(w1>=0)&(w2>=0)
        zz=w0*sz[t,0]+w1*sz[t,1]+w2*sz[t,2]
        sub=dep[ymin:ymax,xmin:xmax]
        m=inside&(zz<sub)
        sub[m]=zz[m]
    return dep
if __name__=='__main__':
    urdf=A+'ur5/tactip/ur5_with_standard_tactip.urdf'
    v,t=visual_meshes_of_link(urdf,'tactip_tip_link')
    # tip frame -> body frame
    Rj=rpy_to_mat([1.57,0,0]); pj=np.array([0,0,0.065])
    vb=v@Rj.T+pj
    cam_pos=np.array([0,0,0.03]); Rc=rpy_to_mat([0,-np.pi/2,np.pi])
    fwd=Rc[:,0]; up=Rc[:,2]; right=np.cross(fwd,up)
    print('fwd',fwd,'up',up,'right',right)
    M=np.stack([right,up,-fwd])  # rows
    vc=(vb-cam_pos)@M.T
    dep=raster(vc[t],128,128,60,0.01,1.0)
    ref=np.load(A+'tactip/reference_images/standard/128x128/nodef_dep.npy')
    b=np.load(A+'tactip/reference_images/standard/128x128/border_mask.npy')
    inner=(b==0)
    d=dep-ref
    print('inner px',inner.sum(),'max abs diff inner',np.abs(d[inner]).max(),'mean',np.abs(d[inner]).mean())
    print('frac inner > 1e-4', (np.abs(d[inner])>1e-4).mean(), ' >1e-5',(np.abs(d[inner])>1e-5).mean())
    for flip in ['none','lr','ud','both','T']:
        r={'none':dep,'lr':dep[:,::-1],'ud':dep[::-1],'both':dep[::-1,::-1],'T':dep.T}[flip]
        print(flip, np.abs((r-ref)[inner]).max(), np.abs((r-ref)[inner]).mean())
    np.save('/root/repo/scratch/skin_dep.npy',dep)
