import sys; sys.path.insert(0,'/root/repo')
import bench
cfg=bench.CONFIGS['cfg4s']
plan,support=bench.make_plan(cfg,50,0,1)
m=bench.create_model(cfg,support,0,1,0,None,use_graph=False)
print(m.get_debug('occ_score_tile',(1,)))
