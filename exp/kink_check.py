import torch, sys
sys.path.insert(0,'/root/repo')
from tests.util import random_graph
from oracle import layers as OL
from het_amd.layers import HET_RGATLayer
from het_amd.backend import rgat_fused_layer as FL
def run(lit, per_edge):
    FL.LITERAL_ER=lit; FL.PER_EDGE=per_edge
    g=random_graph(seed=48, n=260, r=3, e=4000, shuffle=False)
    H,K,X=8,64,64
    torch.manual_seed(0)
    R,N=g.get_num_rels(), g.get_num_nodes()
    layer=HET_RGATLayer(K,X,R,H,bias=True,self_loop=True,dropout=0.0)
    with torch.no_grad(): layer.h_bias.uniform_(-0.1,0.1)
    x=torch.randn(N,K)*0.5; go=torch.randn(N,X)
    s=g.get_separate_coo_original()
    p={n:t.detach().double().requires_grad_(True) for n,t in layer.named_parameters()}
    x64=x.double().requires_grad_(True)
    ref=OL.rgat_layer(x64,p["conv_weights"],p["attn_l"],p["attn_r"],s["rel_ptrs"],s["row_indices"],s["col_indices"],N,0.2,p["loop_weight"],p["h_bias"])
    gx,=torch.autograd.grad(ref,[x64],go.double())
    # min |z|
    W=p["conv_weights"].detach(); 
    rel=torch.repeat_interleave(torch.arange(R), s["rel_ptrs"][1:]-s["rel_ptrs"][:-1])
    fs=torch.einsum('ek,ehkd->ehd', x.double()[s["row_indices"]], W[rel])
    fd=torch.einsum('ek,ehkd->ehd', x.double()[s["col_indices"]], W[rel])
    z=(fs*p["attn_l"].detach()[rel]).sum(-1)+(fd*p["attn_r"].detach()[rel]).sum(-1)
    g.to_("cuda"); layer=layer.to("cuda"); xd=x.cuda().requires_grad_(True)
    out=layer(g,xd); out.backward(go.cuda())
    err=(xd.grad.cpu().double()-gx).abs()
    i=err.argmax()
    print('lit',lit,'per_edge',per_edge,'max err',err.max().item(),'at',divmod(i.item(),K),'min|z|',z.abs().min().item(), 'edges with |z|<1e-6', int((z.abs()<1e-6).sum()))
    zz=z.abs().flatten(); k=zz.argmin(); e=k//H
    print('   edge',e.item(),'src',s["row_indices"][e].item(),'dst',s["col_indices"][e].item())
for lit in (False,True):
    for pe in (False,True):
        run(lit,pe)
