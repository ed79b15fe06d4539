import numpy as np, sys
st=np.load(sys.argv[1]); st=st[st[:,0]>=0]; n=len(st)
life=(st[:,1]-st[:,0])*1e3; wg=np.arange(n)//4; R=(wg.max()+1)//256
print(sys.argv[1], "span %.1f mean life %.1f p10 %.1f p90 %.1f max %.1f"%((st[:,1].max()-st[:,0].min())*1e3, life.mean(), np.percentile(life,10), np.percentile(life,90), life.max()))
print(" life by rank:", [round(float(life[wg//256==k].mean()),1) for k in range(R)])
print(" entries by rank:", [round(float(st[wg//256==k,5].mean())) for k in range(R)])
