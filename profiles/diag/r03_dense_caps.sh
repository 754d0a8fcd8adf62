# dense / HiFi-like shapes: fixed wave shares against the model's pick
for c in 7,6,5 7,5,1 8,4,1 9,3,1 8,5,2 ""; do
  echo "caps '$c'"
  env ${c:+HYPO_POA_CAPS=$c} python profiles/dense_rate.py 2>&1 | grep "dense-SR" | cut -c1-110
  env ${c:+HYPO_POA_CAPS=$c} python profiles/hifi_rate.py 2>&1 | grep "HiFi" | cut -c1-100
done
